#!/bin/bash
# Build ablation variants of the library (timing experiments only; results are WRONG by construction).
# usage: tools/ablate.sh 1 2 4 ...   -> libNeuralAudioCAPI_ablN.so
cd "$(dirname "$0")/../neuralaudio_amd/csrc"
for m in "$@"; do
  make -j8 SUFFIX=_abl$m EXTRA="-DNA_ABL=$m" 2>&1 | grep -E "error" 
done
ls ../libNeuralAudioCAPI_abl*.so
