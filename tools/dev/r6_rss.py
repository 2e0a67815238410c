"""resident set of the tools on small inputs: what the process holds whatever the file"""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "dextractor_amd", "bin")
D = tempfile.mkdtemp(prefix="rss.", dir="/dev/shm")
try:
    import random
    random.seed(1)
    def fasta(path, n, L):
        with open(path, "wb") as f:
            for i in range(n):
                f.write(b">m000_000/%d/0_%d RQ=0.850\n" % (i, L))
                s = bytes(random.choice(b"acgt") for _ in range(80)) * (L // 80)
                for a in range(0, len(s), 80): f.write(s[a:a + 80] + b"\n")
    fasta(os.path.join(D, "t.fasta"), 100, 8000)
    fasta(os.path.join(D, "m.fasta"), 20000, 8000)        # 160 MB
    def rss(cmd, stdin=None, stdout=None, env=None):
        p = subprocess.Popen(cmd, cwd=D, stdin=stdin, stdout=stdout, env=dict(os.environ, **(env or {})))
        _, st, ru = os.wait4(p.pid, 0)
        return st, ru.ru_maxrss / 1024.0
    print("dexta of 0.8 MB file      ", rss([os.path.join(BIN, "dexta"), "-k", "t.fasta"]))
    print("dexta -i of 0.8 MB        ", rss([os.path.join(BIN, "dexta"), "-i"], stdin=open(os.path.join(D, "t.fasta"), "rb"), stdout=open(os.path.join(D, "t2.dexta"), "wb")))
    cat = subprocess.Popen(["cat", os.path.join(D, "m.fasta")], stdout=subprocess.PIPE)
    print("cat 160 MB | dexta -i     ", rss([os.path.join(BIN, "dexta"), "-i"], stdin=cat.stdout, stdout=open(os.path.join(D, "m2.dexta"), "wb")))
    cat.wait()
    print("dexta of 160 MB file      ", rss([os.path.join(BIN, "dexta"), "-k", "m.fasta"]))
    # what of it is the runtime's: a process that opens a context and leaves
    import ctypes
    code = "import ctypes,os,resource; l=ctypes.CDLL(os.path.join(%r,'dextractor_amd','libdexgpu.so')); c=ctypes.c_void_p(); r=l.dx_open(0, ctypes.byref(c)); print('ctx', r, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1024.0, 'MB')" % ROOT
    subprocess.run([sys.executable, "-c", code])
    code2 = "import ctypes,resource; h=ctypes.CDLL('libamdhip64.so'); n=ctypes.c_int(); h.hipGetDeviceCount(ctypes.byref(n)); print('after hipGetDeviceCount', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1024.0); h.hipSetDevice(0); p=ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), 1<<20); print('after a hipMalloc', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1024.0, 'MB')"
    subprocess.run([sys.executable, "-c", code2])
finally:
    shutil.rmtree(D, True)
