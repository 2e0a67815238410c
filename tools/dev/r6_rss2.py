"""where the resident set of `dexta -i` over a pipe goes: /proc/<pid>/status polled, the largest mappings at the peak"""
import os, subprocess, sys, tempfile, shutil, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "dextractor_amd", "bin")
D = tempfile.mkdtemp(prefix="rss.", dir="/dev/shm")
try:
    random.seed(1)
    with open(os.path.join(D, "m.fasta"), "wb") as f:
        for i in range(20000):
            L = 8000
            f.write(b">m000_000/%d/0_%d RQ=0.850\n" % (i, L))
            s = bytes(random.choice(b"acgt") for _ in range(80)) * (L // 80)
            f.write(b"".join(s[a:a + 80] + b"\n" for a in range(0, len(s), 80)))
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for env in ({},):
        cat = subprocess.Popen("for i in $(seq %d); do cat m.fasta; done" % reps, shell=True, cwd=D, stdout=subprocess.PIPE)
        p = subprocess.Popen([os.path.join(BIN, "dexta"), "-i"], cwd=D, stdin=cat.stdout, stdout=open(os.path.join(D, "o.dexta"), "wb"), env=dict(os.environ, **env))
        peak, peak_status, peak_maps = 0, "", ""
        while p.poll() is None:
            try:
                st = open("/proc/%d/status" % p.pid).read()
                rss = int([l for l in st.splitlines() if l.startswith("VmRSS")][0].split()[1])
                if rss > peak * 1.05:
                    peak = rss
                    peak_status = "\n".join(l for l in st.splitlines() if l.startswith(("VmRSS", "RssAnon", "RssFile", "RssShmem", "VmLck", "VmPin")))
                    rows, cur = [], None
                    for l in open("/proc/%d/smaps" % p.pid):
                        if l[0] in "0123456789abcdef" and "-" in l.split()[0]: cur = l.strip()
                        elif l.startswith("Rss:"): rows.append((int(l.split()[1]), cur))
                    rows.sort(reverse=True)
                    peak_maps = "\n".join("%9d kB  %s" % r for r in rows[:14])
            except Exception as e:
                pass
            time.sleep(0.03)
        cat.wait()
        print("==", env, "input", reps * 0.162, "GB, exit", p.returncode, "peak VmRSS kB", peak)
        print(peak_status); print(peak_maps)
finally:
    shutil.rmtree(D, True)
