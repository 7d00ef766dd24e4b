#!/bin/bash
out=gpurun_out/r2_run3; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -30 $out/pytest.log
ab() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/ab_$name.json 2> $out/ab_$name.err; python - <<PY
import json
try:
    d = json.load(open('$out/ab_$name.json')); print('$name', d['ms_per_step'], d['value'])
except Exception as e:
    print('$name FAILED', e)
PY
}
ab base A=1
ab nosavegrad PRISMER_SAVE_ACT_GRAD=0
ab ring4 PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_ring4.so
ab ring6 PRISMER_HIP_LIB=prismer_amd/lib/libprismer_hip_ring6.so
ab nobig PH_GEMM_BIG=0
ab big150 PH_GEMM_BIG_MIN_TILES=150
ab base2 A=1
