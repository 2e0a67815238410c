#!/bin/bash
# end-to-end CLI timing on a 1 GB .quiva in tmpfs
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
for i in 1 2; do DEXGPU_TIMING=1 $R/dextractor_amd/bin/dexqv -k s; done
for i in 1 2; do DEXGPU_TIMING=1 $R/dextractor_amd/bin/undexqv -k s; done
rm -rf $D
