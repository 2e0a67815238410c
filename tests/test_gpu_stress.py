"""Random shapes (needs an MI355X): the two stress tools, a short run each with a fixed seed.

tools/stress_files.py   the six file drivers against the oracle (lengths 0 .. 90 k, run densities 0.02 .. 0.995, lossy,
                        line widths 1 .. 1000): identical bytes, identical refusals
tools/stress_decode.py  encode with the group index -> decode on the device, whole batches and random contiguous parts:
                        the text that went in, nothing written outside the part
tools/stress_cli.py     malformed text inputs through the packing tools and the reference's own binaries: exit code, message, output
                        (found: the message for a last line without a newline, dexar's "Fasta line", a header with the file's end behind it)
The first two found what the hand-made cases had not: files with a stream of ONE symbol (its Huffman code has no bits), which the
reference writes and cannot read back -- dx_qv_decode used to return a text with zeros in it and now refuses them."""
import os
import subprocess
import sys

import pytest

from _flags import set_flag, test_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,rounds,seed", [("stress_files.py", 60, 5), ("stress_decode.py", 80, 11), ("stress_cli.py", 40, 3)])
def test_random_shapes(tool, rounds, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(rounds), str(seed)], capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-2000:])
    assert b"OK" in r.stdout.splitlines()[-1]


def test_random_shapes_with_the_records_walked_on_the_device():
    """The same file-driver rounds with undexqv's record walk on the device whatever the file's size (the product asks for
    256 MB), in pieces of 4 KiB: escape schemes, files without a run character, records longer than a piece, ..."""
    env = dict(os.environ, DEXGPU_TEST=test_env(device_walk_min=0, walk_piece=4096))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_files.py"), "60", "23"], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-2000:])
    assert b"OK" in r.stdout.splitlines()[-1]
