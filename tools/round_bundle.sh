#!/bin/bash
# Round-end verification + measurement bundle for ONE gpurun call (run from the repo root on the GPU box):
#   gpurun --timeout 2000 -- 'bash tools/round_bundle.sh'
# Outputs land in gpurun_out/; copy them to profiles/rNN_* afterwards (see tools/README.md).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -1 gpurun_out/bench_full.json | cut -c1-250
python tools/bench_models.py --steps 5 > gpurun_out/models.jsonl 2>/dev/null
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_final" -o final -- python "$REPO/bench.py" --headline-only --steps 20 > "$REPO/gpurun_out/prof_final.log" 2>&1
i=0
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$REPO/gpurun_out/pmcf_$i" -- python "$REPO/bench.py" --precision f16 --steps 5 --warmup 2 --headline-only > /dev/null 2>&1
done
cd "$REPO"
python tools/rocprof_summary.py "$(ls gpurun_out/prof_final/*.db | head -1)" > gpurun_out/final_kernel_stats.md
python tools/pmc_traffic.py f16 "gemm_f16_dma_kernel<128, 128|gemm_f16_p8_kernel" gpurun_out/pmcf_1 gpurun_out/pmcf_2 gpurun_out/pmcf_3 > gpurun_out/pmc_f16.json
head -8 gpurun_out/final_kernel_stats.md | cut -c1-160
