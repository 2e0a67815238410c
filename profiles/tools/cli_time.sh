#!/bin/bash
# end-to-end CLI timing on a 1 GB .quiva in tmpfs; undexqv with and without the walk's group index
R=$PWD
D=/dev/shm/clit; rm -rf $D; mkdir -p $D
python3 - <<PY
import sys; sys.path.insert(0,'$R')
from dextractor_amd import synth
c = synth.make_quiva(20000, seed=5, mean=10000)
open('$D/s.quiva','wb').write(c.text)
PY
ls -la $D
cd $D
T() { local a=$(date +%s%N); "$@"; local b=$(date +%s%N); echo "   wall $(( (b - a) / 1000000 )) ms: $*"; }
for i in 1 2 3; do DEXGPU_TIMING=1 T $R/dextractor_amd/bin/dexqv -k s; done
cp s.quiva s0.quiva
echo "== undexqv -U with the walk's group index (DEXGPU_TEST=walk_index)"
for i in 1 2 3; do DEXGPU_TEST=walk_index DEXGPU_TIMING=1 T $R/dextractor_amd/bin/undexqv -k -U s; cmp s.quiva s0.quiva && echo "   identical to the input"; done
echo "== undexqv -U without (the default)"
for i in 1 2 3; do DEXGPU_TIMING=1 T $R/dextractor_amd/bin/undexqv -k -U s; cmp s.quiva s0.quiva && echo "   identical to the input"; done
rm -rf $D
