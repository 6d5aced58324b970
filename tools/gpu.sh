#!/bin/bash
# The one GPU-box driver script (run through gpurun from the repo root):   bash tools/gpu.sh <task> [<task> ...]
# Tasks run in the order given and stop at the first failing one.  Everything logs under gpurun_out/ (merged back by gpurun);
# what should be judged is copied from there into profiles/ by hand.  rocprofv3 always gets the python program directly after `--`.
#   tests        python -m pytest tests -m gpu            -> gpurun_out/pytest_gpu.log            (PYTEST_ARGS / PYTEST_PATHS to narrow)
#   bench        python bench.py (default: global batch 8192; BENCH_ARGS="--local-batch 1024" = the 1024-pair shard) -> gpurun_out/bench.json / bench.err    (BENCH_ARGS, STEPS)
#   prof         rocprofv3 --kernel-trace --stats of bench.py (3 steps)          -> gpurun_out/prof/bench_kernel_stats.csv
#   replay       rocprofv3 --kernel-trace --stats of tools/gemm_replay.py (ONE stream: bench.py's roofline launch set)
#                                                         -> gpurun_out/replay/replay_kernel_stats.csv + per-shape table
#   gemm         tools/gemm_bench.py per-shape TF/s       -> gpurun_out/gemm_shapes.txt
#   pmc_gemm     SQ counter passes (MFMA busy ...) on the NT / TN kernels, three shapes each -> gpurun_out/pmc/gemm_counters.txt
#   pmc_traffic  FETCH_SIZE / WRITE_SIZE passes over the replay launch set   -> gpurun_out/gemm_traffic.json
#   pmc_loss     FETCH_SIZE / WRITE_SIZE + kernel times of the loss head at B = 8192 -> gpurun_out/pmc/loss_counters.txt
#   pmc_bn       tools/bn_one.py: BatchNorm + implicit-convolution kernels at RN50's first-stage shape: times, FETCH_SIZE / WRITE_SIZE
#                                                         -> gpurun_out/bn_conv_times.txt, pmc/bn_conv_counters.txt
#   stamps       tools/gemm_stamps.py: s_memtime stamps of the persistent NT kernel's diagnostic instances (per tile: wait / loop / epilogue) -> gpurun_out/gemm_tile_stamps.txt
#   loss         tools/loss_bench.py                       -> gpurun_out/loss_head_times.txt
#   attn / pmc_attn   tools/attn_bench.py timings / SQ counters of the attention kernels -> gpurun_out/attention_times.txt, pmc/attn_counters.txt
#   configs      bench.py for BASELINE configs C2 / C3 / C5 (per-GPU shards), --precision fp32 and the YAMLs' own RN50 at batch 256
#                                                         -> gpurun_out/bench_c2|c3|c5|fp32|rn50.json
#   corun        tools/corun_bench.py: the persistent NT GEMM with 16 / 32 / 64 CUs taken by another stream, fixed lists vs tile tickets
#                                                         -> gpurun_out/gemm_corun.txt
#   tn_group     tools/tn_group_bench.py: the four weight gradients of a block as one grouped TN launch at the step's block shapes -> gpurun_out/tn_group_bench.txt
#   epi_half     tools/epi_half_chip.py: stamped tiles of the NT epilogues with 0 / 64 / 128 / 192 CUs held by a diagnostic kernel -> gpurun_out/epi_half_chip.txt
#   pmc_pair     tools/pmc_loss_pair.sh: SQ counter passes on the loss head's pairwise kernels at B = 8192 -> gpurun_out/pmc/loss_pair_counters.txt
#   timeline     tools/step_timeline.py over the trace of the last `prof` -> gpurun_out/step_timeline.txt
#   ln           tools/ln_bench.py stand-alone LayerNorm forward / backward rates -> gpurun_out/layernorm_times.txt
#   pipeline     tools/pipeline_bench.py: the input stage per phase (host assembly, H2D, loader alone, the step fed by the loader) -> gpurun_out/pipeline_bench.txt
#   dp2s         the same rehearsal in the default strong-scaling form (global batch split over the ranks, shards by micro-batches) -> gpurun_out/dp2s.json
#   dp2          python bench.py --gpus 2 under SC_DIST_BACKEND=gloo on the one GPU (self-launch rehearsal) -> gpurun_out/dp2.json
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof $R/gpurun_out/pmc $R/gpurun_out/replay
export TMPDIR=/tmp
PMC() { # name, counters..., -- program args
  name=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  (cd /tmp && timeout -k 10 ${PMC_TIMEOUT:-400} rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o p -- python3 "$@" > $R/gpurun_out/pmc/$name.log 2>&1)
  rc=$?; echo "pmc $name rc=$rc"; return $rc
}
for task in "$@"; do
  cd $R
  case $task in
    tests)
      timeout -k 10 ${TEST_TIMEOUT:-1100} python -m pytest ${PYTEST_PATHS:-tests} -m gpu -q --timeout 600 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
      echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20
      [ $rc = 0 ] || exit $rc ;;
    bench)
      timeout -k 10 600 python bench.py --steps ${STEPS:-8} --warmup 2 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
      echo "bench rc=$rc"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
      [ $rc = 0 ] || exit $rc ;;
    dp2)
      SC_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --local-batch ${DP2_BATCH:-256} > gpurun_out/dp2.json 2> gpurun_out/dp2.err; rc=$?
      echo "dp2 rc=$rc"; tail -3 gpurun_out/dp2.err; cat gpurun_out/dp2.json
      [ $rc = 0 ] || exit $rc ;;
    dp2s)   # the default (strong-scaling) form with two ranks: each runs its 1024-pair shard by micro-batches of 512 through Trainer.step_cached
      SC_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --global-batch 2048 --local-batch 512 --steps 2 --warmup 1 > gpurun_out/dp2s.json 2> gpurun_out/dp2s.err; rc=$?
      echo "dp2s rc=$rc"; tail -3 gpurun_out/dp2s.err; cut -c1-400 gpurun_out/dp2s.json
      [ $rc = 0 ] || exit $rc ;;
    prof)
      (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --simulate-dp 1 --global-batch-one-gpu 0 ${BENCH_ARGS} > $R/gpurun_out/rocprof.log 2>&1); rc=$?
      echo "rocprof rc=$rc"
      find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
      python3 tools/kernel_stats.py $R/gpurun_out/prof/bench_kernel_stats.csv 24
      [ $rc = 0 ] || exit $rc ;;
    replay)
      (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/replay -o replay -- python3 $R/tools/gemm_replay.py > $R/gpurun_out/replay.log 2>&1); rc=$?
      echo "replay rc=$rc"
      python3 tools/kernel_stats.py $R/gpurun_out/replay/replay_kernel_stats.csv 10
      python3 tools/replay_table.py $R/gpurun_out/replay/replay_kernel_trace.csv > $R/gpurun_out/replay/replay_per_launch.txt; cat $R/gpurun_out/replay/replay_per_launch.txt
      find $R/gpurun_out/replay -name "*kernel_trace.csv" -size +20M -delete
      [ $rc = 0 ] || exit $rc ;;
    gemm)
      timeout -k 10 600 python tools/gemm_bench.py > gpurun_out/gemm_shapes.txt 2>&1; rc=$?; cat gpurun_out/gemm_shapes.txt; [ $rc = 0 ] || exit $rc ;;
    stamps)
      timeout -k 10 300 python tools/gemm_stamps.py > gpurun_out/gemm_tile_stamps.txt 2>&1; rc=$?; cut -c1-330 gpurun_out/gemm_tile_stamps.txt; [ $rc = 0 ] || exit $rc ;;
    loss)
      timeout -k 10 600 python tools/loss_bench.py > gpurun_out/loss_head_times.txt 2>&1; rc=$?; cat gpurun_out/loss_head_times.txt; [ $rc = 0 ] || exit $rc ;;
    pmc_gemm)
      PMC g_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -- $R/tools/gemm_one.py || exit 1
      PMC g_sq2 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -- $R/tools/gemm_one.py || exit 1
      python3 tools/pmc_table.py gemm_bf16 $R/gpurun_out/pmc/g_sq1 $R/gpurun_out/pmc/g_sq2 > $R/gpurun_out/pmc/gemm_counters.txt; cat $R/gpurun_out/pmc/gemm_counters.txt ;;
    pmc_traffic)
      PMC t_fetch FETCH_SIZE -- $R/tools/gemm_replay.py || exit 1
      PMC t_write WRITE_SIZE -- $R/tools/gemm_replay.py || exit 1
      python3 tools/pmc_traffic.py $R/gpurun_out/pmc/t_fetch $R/gpurun_out/pmc/t_write > $R/gpurun_out/gemm_traffic.json; cat $R/gpurun_out/gemm_traffic.json ;;
    pmc_loss)
      PMC l_fetch FETCH_SIZE -- $R/tools/loss_one.py || exit 1
      PMC l_write WRITE_SIZE GRBM_GUI_ACTIVE -- $R/tools/loss_one.py || exit 1
      python3 tools/pmc_table.py "" $R/gpurun_out/pmc/l_fetch $R/gpurun_out/pmc/l_write > $R/gpurun_out/pmc/loss_counters.txt; cat $R/gpurun_out/pmc/loss_counters.txt ;;
    pmc_bn)
      timeout -k 10 300 python tools/bn_one.py > gpurun_out/bn_conv_times.txt 2>&1 || { cat gpurun_out/bn_conv_times.txt; exit 1; }
      cat gpurun_out/bn_conv_times.txt
      PMC b_fetch FETCH_SIZE -- $R/tools/bn_one.py || exit 1
      PMC b_write WRITE_SIZE GRBM_GUI_ACTIVE -- $R/tools/bn_one.py || exit 1
      python3 tools/pmc_table.py "" $R/gpurun_out/pmc/b_fetch $R/gpurun_out/pmc/b_write | grep -E "bn_|gemm_bf16|splitk" > $R/gpurun_out/pmc/bn_conv_counters.txt; cat $R/gpurun_out/pmc/bn_conv_counters.txt ;;
    pmc_attn)
      PMC a_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $R/tools/attn_bench.py || exit 1
      PMC a_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES -- $R/tools/attn_bench.py || exit 1
      python3 tools/pmc_table.py attn $R/gpurun_out/pmc/a_sq1 $R/gpurun_out/pmc/a_sq2 > $R/gpurun_out/pmc/attn_counters.txt; cat $R/gpurun_out/pmc/attn_counters.txt ;;
    configs)   # one-GPU shard lines of the other BASELINE configurations and the fp32 parity path
      for spec in "c2 --experiment experiment_2 --local-batch 512" "c3 --experiment experiment_4 --local-batch 512" \
                  "c5 --experiment experiment_10 --model ViT-L-14 --local-batch 512" "fp32 --precision fp32 --local-batch 256" \
                  "rn50 --model RN50 --local-batch 256"; do
        set -- $spec; name=$1; shift
        timeout -k 10 500 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --simulate-dp 1 --global-batch-one-gpu 0 "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; rc=$?
        echo "$name rc=$rc"; cut -c1-330 gpurun_out/bench_$name.json; echo; [ $rc = 0 ] || exit $rc
      done ;;
    corun)
      timeout -k 10 300 python tools/corun_bench.py > gpurun_out/gemm_corun.txt 2>&1; rc=$?; cat gpurun_out/gemm_corun.txt; [ $rc = 0 ] || exit $rc ;;
    ln)
      timeout -k 10 300 python tools/ln_bench.py > gpurun_out/layernorm_times.txt 2>&1; rc=$?; cat gpurun_out/layernorm_times.txt; [ $rc = 0 ] || exit $rc ;;
    pipeline)
      timeout -k 10 400 python tools/pipeline_bench.py > gpurun_out/pipeline_bench.txt 2>&1; rc=$?; cat gpurun_out/pipeline_bench.txt; [ $rc = 0 ] || exit $rc ;;
    tn_group)
      timeout -k 10 300 python tools/tn_group_bench.py > gpurun_out/tn_group_bench.txt 2>&1; rc=$?; cat gpurun_out/tn_group_bench.txt; [ $rc = 0 ] || exit $rc ;;
    epi_half)
      timeout -k 10 400 python tools/epi_half_chip.py > gpurun_out/epi_half_chip.txt 2>&1; rc=$?; cat gpurun_out/epi_half_chip.txt; [ $rc = 0 ] || exit $rc ;;
    pmc_pair)
      bash tools/pmc_loss_pair.sh || exit 1 ;;
    timeline)
      python3 tools/step_timeline.py gpurun_out/prof/bench_kernel_trace.csv > gpurun_out/step_timeline.txt; rc=$?; cat gpurun_out/step_timeline.txt; [ $rc = 0 ] || exit $rc ;;
    attn)
      timeout -k 10 300 python tools/attn_bench.py > gpurun_out/attention_times.txt 2>&1; rc=$?; cat gpurun_out/attention_times.txt; [ $rc = 0 ] || exit $rc ;;
    *) echo "unknown task $task"; exit 2 ;;
  esac
  find $R/gpurun_out/pmc -name "*.csv" -size +8M -delete 2>/dev/null
done
