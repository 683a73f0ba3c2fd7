#!/bin/bash
# round 2, GPU call F: the whole GPU suite on the tuned build, smoke(), default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -120 ) > $O/r02_f_gpu_tests.log
tail -6 $O/r02_f_gpu_tests.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $O/r02_f_smoke.log
cat $O/r02_f_smoke.log
( time timeout 1200 python bench.py ) > $O/r02_f_bench_default.json 2> $O/r02_f_bench_default.err
tail -3 $O/r02_f_bench_default.err
cut -c1-400 $O/r02_f_bench_default.json
