#!/bin/bash
# Round-5 evidence run (on the GPU box, through gpurun): PMC passes + bench lines of every workload on the final kernels.
#   gpurun --timeout 1200 -- 'bash tools/r05_profiles.sh pmc1'   (sponza F=20 / F=64)
#   ... pmc2 (dragon, glass), pmc3 (dragon8m), bench (all N = 1 lines + the N = 8 / N = 4 rehearsals)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r05p
mkdir -p $OUT
case "$1" in
  pmc1) python3 tools/pmc_traffic.py --workload sponza --frames 20 --out $OUT > $OUT/pmc_sponza_f20.log 2>&1
        python3 tools/pmc_traffic.py --workload sponza --frames 64 --out $OUT > $OUT/pmc_sponza_f64.log 2>&1 ;;
  pmc2) python3 tools/pmc_traffic.py --workload dragon --frames 64 --out $OUT > $OUT/pmc_dragon_f64.log 2>&1
        python3 tools/pmc_traffic.py --workload glass --frames 64 --out $OUT > $OUT/pmc_glass_f64.log 2>&1 ;;
  pmc3) python3 tools/pmc_traffic.py --workload dragon8m --frames 64 --out $OUT > $OUT/pmc_dragon8m_f64.log 2>&1 ;;
  bench)
    python3 bench.py --steps 20 --warmup 5 > $OUT/bench_sponza_steps20.json 2> $OUT/bench_sponza_steps20.err
    python3 bench.py > $OUT/bench_sponza_steps128.json 2> $OUT/bench_sponza_steps128.err
    for wl in sponza4k dragon dragon80k dragon8m glass spheres; do
      python3 bench.py --workload $wl --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
    done
    python3 bench.py --gpus 8 --steps 20 --warmup 5 > $OUT/bench_n8_rehearsal.json 2> $OUT/bench_n8_rehearsal.err
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --steps 20 --warmup 5 \
        > $OUT/bench_n4_torchrun_rehearsal.json 2> $OUT/bench_n4_torchrun_rehearsal.err
    python3 tools/rank_share.py --library-split --steps 20 128 --ranks 8 --out $OUT/rank_share.json > $OUT/rank_share.txt 2>&1 ;;
esac
ls -la $OUT | tail -40
