"""dx_host.c (indexers, .dexqv boundary walk, table builder) compiled with AddressSanitizer + UBSan
and driven over the golden files plus truncated and bit-damaged copies of them: no out-of-bounds
access, whatever the bytes.  CPU only."""
import glob
import gzip
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_host_code_under_asan(tmp_path):
    exe = str(tmp_path / "host_asan")
    cc = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
          "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_asan", "driver.c"),
          os.path.join(ROOT, "dextractor_amd", "csrc", "dx_host.c"), "-o", exe, "-lm"]
    r = subprocess.run(cc, capture_output=True)
    if r.returncode != 0 and b"sanitize" in r.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert r.returncode == 0, r.stderr.decode()
    files = []
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*"))):
        ext = p[:-3].rsplit(".", 1)[-1] if p.endswith(".gz") else p.rsplit(".", 1)[-1]
        if ext not in ("quiva", "fasta", "arrow", "dexqv"):
            continue
        if p.endswith(".gz"):
            q = str(tmp_path / os.path.basename(p)[:-3])
            with gzip.open(p, "rb") as f, open(q, "wb") as g:
                g.write(f.read())
            p = q
        files.append(p)
    assert len(files) >= 15
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + files, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:] + r.stderr[-6000:]).decode(errors="replace")
    assert r.stdout.startswith(b"ok ")
