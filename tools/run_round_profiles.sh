#!/bin/bash
# The round's measurement artefacts, on the GPU box: the bench line (+ full record), and rocprofv3 --kernel-trace --stats
# summaries of (a) the headline command, (b) the default batched path, (c) the fp32-MFMA batched path.
# (60 timed launches each: after an idle gap the clocks take ~8 launches to come up — k_mfma_filter 3.6 -> 3.03 ms, tools/kcalls.sh)
# bash tools/run_round_profiles.sh r03
tag=${1:-rXX}; root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out; cd $root
python3 bench.py --gpus 1 --steps 100 --warmup 10 > gpurun_out/${tag}_bench_n1.jsonl 2> gpurun_out/${tag}_bench_n1.err
cp gpurun_out/bench_full.json gpurun_out/${tag}_bench_full.json 2>/dev/null
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/rp_h && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_h -o h -- python3 $root/bench.py --steps 100 --warmup 10 --no-also --no-cpu-baseline > $root/gpurun_out/${tag}_bench_under_rocprof_10Mx768.jsonl 2>/dev/null)
cp $(find /tmp/rp_h -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_10Mx768_kernel_stats.csv 2>/dev/null
for f in "" fp32; do
  (cd /tmp && rm -rf /tmp/rp_b && DEV_REPS=60 DEV_FILTER=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_b -o b -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1)
  cp $(find /tmp/rp_b -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_batched_${f:-default}_256x1Mx768_kernel_stats.csv 2>/dev/null
done
(cd /tmp && rm -rf /tmp/rp_b && DEV_REPS=60 DEV_BF16_ROWS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_b -o b -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1)
cp $(find /tmp/rp_b -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_batched_bf16rows_256x1Mx768_kernel_stats.csv 2>/dev/null
wc -c gpurun_out/${tag}_bench_n1.jsonl; head -c 3500 gpurun_out/${tag}_bench_n1.jsonl; echo; head -5 gpurun_out/${tag}_10Mx768_kernel_stats.csv; head -4 gpurun_out/${tag}_batched_default_256x1Mx768_kernel_stats.csv; head -3 gpurun_out/${tag}_batched_fp32_256x1Mx768_kernel_stats.csv
# round 5: configs[1] (one query over 1M x 768, single launch per query) under rocprofv3, the concurrent-caller bench, the fp32-MFMA filter's
# matrix-instruction counters, the sharded handle's batch path (8 co-located shards: device-side redo, no host round trip per shard)
(cd /tmp && rm -rf /tmp/rp_1m && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_1m -o s -- python3 $root/tools/dev_scan_1m.py > $root/gpurun_out/${tag}_1Mx768_under_rocprof.txt 2>/dev/null)
cp $(find /tmp/rp_1m -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_1Mx768_kernel_stats.csv 2>/dev/null
python3 tools/dev_scan_1m.py > gpurun_out/${tag}_1Mx768_scan.txt 2>/dev/null
QV_SCAN_FUSE=0 python3 tools/dev_scan_1m.py >> gpurun_out/${tag}_1Mx768_scan.txt 2>/dev/null
python3 tools/bench_callers.py --out gpurun_out/${tag}_callers.json > gpurun_out/${tag}_callers.log 2>&1
QV_COALESCE=0 python3 tools/bench_callers.py --flat-callers 1,8,64 --graph-callers 1,64 --out gpurun_out/${tag}_callers_off.json > gpurun_out/${tag}_callers_off.log 2>&1
DEV_FILTER=fp32 DEV_REPS=30 bash tools/pmc_kernel.sh k_mfma_filter gpurun_out/${tag}_mfma_pmc.txt "SQ_INSTS_VALU_MFMA_MOPS_F32/SQ_INSTS_VALU_MFMA_F32/SQ_VALU_MFMA_BUSY_CYCLES/SQ_BUSY_CYCLES/SQ_INSTS_MFMA/SQ_INSTS_VALU/GRBM_GUI_ACTIVE/SQ_WAVES/SQ_BUSY_CU_CYCLES" -- python3 $root/tools/dev_batched.py cosine 256 1000000 768 10 > /dev/null 2>&1
python3 tools/dev_sharded_batch.py 8 256 8000000 > gpurun_out/${tag}_sharded_batch.txt 2>&1
head -3 gpurun_out/${tag}_1Mx768_kernel_stats.csv; cat gpurun_out/${tag}_1Mx768_scan.txt; cat gpurun_out/${tag}_mfma_pmc.txt; tail -5 gpurun_out/${tag}_sharded_batch.txt; cut -c1-260 gpurun_out/${tag}_callers.log
# round 5 (late): the 1M x 768 MaxLevel=1 graph — build and device-resident / host-pointer search under rocprofv3 (wave pass, exact-heap redo,
# link phase by kernel), and the lone traversal's phases (-DQV_HNSW_PROF build of qv_hnsw.hip, when tools/build_variant.sh has made one)
(cd /tmp && rm -rf /tmp/rp_g && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_g -o g -- python3 $root/tools/dev_hnsw_tput.py 64,128,256 > $root/gpurun_out/${tag}_hnsw_1Mx768_under_rocprof.txt 2>/dev/null)
cp $(find /tmp/rp_g -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_hnsw_build_search_1Mx768_kernel_stats.csv 2>/dev/null
python3 tools/dev_hnsw_tput.py 64,128,256,512 2>/dev/null | grep -E "^build|^ef" > gpurun_out/${tag}_hnsw_1Mx768.txt
if [ -f quiver_amd/lib/libqv_prof.so ]; then QV_LIB_PATH=$root/quiver_amd/lib/libqv_prof.so python3 tools/dev_hnsw_phase.py 1000000 4 2>/dev/null | grep -A30 -e "---- search" | grep -E "^blk|^ef|^lat" | tail -3 > gpurun_out/${tag}_hnsw_lone_phases.txt; fi
cat gpurun_out/${tag}_hnsw_1Mx768.txt; head -8 gpurun_out/${tag}_hnsw_build_search_1Mx768_kernel_stats.csv | cut -c1-200
# round 5 (late): one query per call on short collections (k_flat_scan_split) under rocprofv3, and the shared-pass form by queries per call
(cd /tmp && rm -rf /tmp/rp_s && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_s -o s -- python3 $root/tools/dev_mid_latency.py 768 10000,30000,100000 > $root/gpurun_out/${tag}_short_768_under_rocprof.txt 2>/dev/null)
cp $(find /tmp/rp_s -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_short_768_kernel_stats.csv 2>/dev/null
python3 tools/dev_mid_latency.py 768 128,5000,10000,30000,60000,100000,160000 2>/dev/null | grep "^rows" > gpurun_out/${tag}_short_latency.txt
python3 tools/dev_mid_latency.py 1536 10000,30000 2>/dev/null | grep "^rows" >> gpurun_out/${tag}_short_latency.txt
python3 tools/dev_mid_latency.py 256 10000,30000 2>/dev/null | grep "^rows" >> gpurun_out/${tag}_short_latency.txt
python3 tools/dev_split_mq.py 768 2>/dev/null | grep "^rows" >> gpurun_out/${tag}_short_latency.txt
cat gpurun_out/${tag}_short_latency.txt; head -4 gpurun_out/${tag}_short_768_kernel_stats.csv | cut -c1-200
