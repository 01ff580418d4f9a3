#!/bin/bash
# Every profile the DESIGN.md numbers come from, in one gpurun call.  Output: gpurun_out/profiles_new/ (copy into profiles/).
#   1. the default bench command: --kernel-trace --stats, then separate --pmc passes (tools/prof_bench.sh)
#   2. kernel traces of the secondary workloads: C3 batch, C4 + streaming, the full registration (rotation search,
#      pre-shape, pose application, AIVS), written as one markdown file
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof_bench.sh $TAG > gpurun_out/prof_bench_$TAG.txt 2>&1 || { echo "prof_bench failed"; tail -5 gpurun_out/prof_bench_$TAG.txt; exit 1; }
OUT=gpurun_out/profiles_new/${TAG}_secondary_kernels.md
echo "# rocprofv3 kernel traces of the secondary workloads ($TAG)" > $OUT
echo >> $OUT
run() {   # dir title cmd...
  local d=$1 title=$2; shift 2
  rm -rf gpurun_out/$d
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$d -- "$@" > gpurun_out/$d.txt 2>&1 || { echo "$d failed"; tail -3 gpurun_out/$d.txt; return 1; }
  python3 tools/trace_table.py gpurun_out/$d "$title" "$*" >> $OUT
  echo '```' >> $OUT; grep -v amdgpu.ids gpurun_out/$d.txt | grep -v "rocprofv3\]" | tail -4 | cut -c1-1200 >> $OUT; echo '```' >> $OUT; echo >> $OUT
}
run pc3 "C3: 1024 x (10k x 10k) batch, 20 iterations + fitness, cell lists and brute force (2 calls each)" python3 tools/bench_configs.py c3 &&
run pc4 "C4 (1M x 1M: pre-shape, NN pass, 10 iterations) and the streaming kernels at 64M / 16M points" python3 tools/bench_configs.py c4 stream &&
run preg "Full registration of the reference's Bunny pair: AIVS x2, pre-shape, 729-candidate rotation search, candidate ICP batch, full-cloud transform" python3 tools/register_time.py
echo "rc=$?"; ls gpurun_out/profiles_new
