#!/bin/bash
# One profiling pass of the round on the GPU box: kernel trace of the bench, PMC traffic of the step's kernels, MFMA busy of the
# window-attention kernels.  Output under gpurun_out/prof/ ; summaries are copied into profiles/ by hand.
# (rocprofv3: the program itself after --, counters in their own passes, never together with trace domains other than kernel-trace)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-steps 0 --sustain-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/bench_under_trace.json 2> $O/kt.err
python3 tools/prof_summary.py $(ls $O/kt/*/*kernel_stats.csv | head -1) 25 60 > $O/kernel_stats_summary.txt
python3 tools/trace_steps.py $(ls $O/kt/*/*kernel_trace.csv | head -1) 10 60 > $O/step_breakdown.txt 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
echo "kernel trace done"
# counter passes on the benched step launched from Python (--no-capture: a replayed hipGraph under counter collection does not finish in reasonable time)
PMCBENCH="python3 bench.py --no-capture --steps 2 --warmup 1 --no-cpu-baseline --kernel-steps 0 --sustain-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $PMCBENCH > /dev/null 2> $O/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $PMCBENCH > /dev/null 2> $O/write.err
echo "write pass done"
python3 tools/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/pmc_traffic.json > /dev/null
for m in infer train; do
  echo "fused $m"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/mf_$m -- python3 tools/pmc_fused.py $m 6 > /dev/null 2> $O/mf_$m.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -- python3 tools/pmc_fused.py $m 6 > /dev/null 2> $O/kt_$m.err
done
for C in 192 384; do
  echo "qkv + attention fused, C = $C"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/mf_qa$C -- python3 tools/pmc_qkv_attn.py $C 6 > /dev/null 2> $O/mf_qa$C.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_qa$C -- python3 tools/pmc_qkv_attn.py $C 6 > /dev/null 2> $O/kt_qa$C.err
done
echo "attention kernels"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --output-format csv -d $O/mf_attn -- python3 tools/pmc_attn.py > /dev/null 2> $O/mf_attn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_attn -- python3 tools/pmc_attn.py > /dev/null 2> $O/kt_attn.err
python3 tools/pmc_mfma_summary.py $(ls $O/mf_infer/*/*counter_collection.csv | head -1) $(ls $O/kt_infer/*/*kernel_stats.csv | head -1) $O/window_attention_mfma.json \
    $(ls $O/mf_train/*/*counter_collection.csv | head -1) $(ls $O/kt_train/*/*kernel_stats.csv | head -1) \
    $(ls $O/mf_attn/*/*counter_collection.csv | head -1) $(ls $O/kt_attn/*/*kernel_stats.csv | head -1) \
    $(ls $O/mf_qa192/*/*counter_collection.csv | head -1) $(ls $O/kt_qa192/*/*kernel_stats.csv | head -1) \
    $(ls $O/mf_qa384/*/*counter_collection.csv | head -1) $(ls $O/kt_qa384/*/*kernel_stats.csv | head -1) > /dev/null
for m in infer train attn qa192 qa384; do python3 tools/pmc_rows.py $(ls $O/mf_$m/*/*counter_collection.csv | head -1) > $O/sq_counters_$m.txt; done
# keep the merge small: raw traces are large
rm -rf $O/kt $O/fetch $O/write $O/mf_* $O/kt_*
ls -la $O
head -30 $O/kernel_stats_summary.txt
cat $O/window_attention_mfma.json
