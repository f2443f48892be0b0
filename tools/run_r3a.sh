set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 180 --timeout-method thread -k "visit or extend_kernel or incoherent or spill or variant or work_counters" > gpurun_out/r3a/tests.txt 2>&1; rc=$?; tail -5 gpurun_out/r3a/tests.txt; [ $rc -eq 0 ] || exit $rc
tools/ab_bench.sh cuda_path_tracer_amd/ab/libcpt_amd_r02.so default 2>&1 | tee gpurun_out/r3a/ab.txt
