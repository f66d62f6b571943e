#!/bin/bash
# Round-end verification + measurement bundle for ONE gpurun call (run from the repo root on the GPU box):
#   gpurun --timeout 2400 -- 'bash tools/round_bundle.sh r04'
# Everything lands in gpurun_out/<tag>_*; tools/collect_profiles.sh <tag> copies the judged artefacts to profiles/.
# An optional second argument selects parts (default: all): "tests bench models prof pmc" -- several short gpurun
# calls lose less than one long one when a box is lost.
TAG=${1:-r04}
PARTS=${2:-tests bench models prof pmc}
want() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
if want tests && [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short > "$OUT/${TAG}_pytest.log" 2>&1; tail -3 "$OUT/${TAG}_pytest.log"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
if want bench; then
# the driver's line (default: ECAPA-512, fp32 headline + both fast modes + config 1 + PLDA + cpu baseline)
# (the driver's command line; stdout = the compact line, bench_detail.json = the full record)
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/${TAG}_bench_n1.err"; cut -c1-200 "$OUT/${TAG}_bench_n1.json"; wc -c "$OUT/${TAG}_bench_n1.json"
cp bench_detail.json "$OUT/${TAG}_bench_detail.json" 2>/dev/null
fi
if want models; then
# the other families of BASELINE.json (configs 2-3) through the same bench.py
rm -f "$OUT/${TAG}_bench_models.jsonl" "$OUT/${TAG}_bench_sets.jsonl"
for m in ECAPA_TDNN_GLOB_c1024 ResNet34 ResNet221 CAMPPlus; do
  python bench.py --model $m --steps 10 --windows 3 --cpu-utts 300 --no-configs 2> /dev/null | tail -1 >> "$OUT/${TAG}_bench_models.jsonl"
done
# BASELINE configs 2 / 3 as fixed-size sets (strong scaling; at N = 1 here, `--gpus N` on a multi-GPU node)
python bench.py --workload vox1o --steps 3 --warmup 1 2> /dev/null | tail -1 >> "$OUT/${TAG}_bench_sets.jsonl"
python bench.py --workload vox1o --model ResNet221 --steps 2 --warmup 1 2> /dev/null | tail -1 >> "$OUT/${TAG}_bench_sets.jsonl"
python bench.py --workload stream10k --steps 3 --warmup 1 2> /dev/null | tail -1 >> "$OUT/${TAG}_bench_sets.jsonl"
# a ragged batch (lengths within 12 % of each other) against the uniform batch of the same size, per family
rm -f "$OUT/${TAG}_ragged.jsonl"
for m in ECAPA_TDNN_GLOB_c512 ECAPA_TDNN_GLOB_c1024 ResNet34 ResNet221 CAMPPlus; do
  b=256; case $m in ResNet34|CAMPPlus) b=512;; esac
  timeout 600 python tools/bench_ragged.py --model $m --batch $b --steps 4 2> /dev/null | tail -1 >> "$OUT/${TAG}_ragged.jsonl"
done
# end to end from wave files in /dev/shm through the batch driver (one engine, two lanes, f16)
timeout 600 python tools/bench_driver.py 2> /dev/null | tail -1 > "$OUT/${TAG}_driver.jsonl"; cut -c1-300 "$OUT/${TAG}_driver.jsonl"
# ... and a list of 1 024 whole utterances of 4 - 12 s (what a VoxCeleb-like test set looks like)
timeout 600 python tools/bench_driver.py --long_n 1024 2> /dev/null | tail -1 > "$OUT/${TAG}_driver_long.jsonl"; cut -c1-300 "$OUT/${TAG}_driver_long.jsonl"
fi
cd /tmp && export TMPDIR=/tmp
# kernel tables: one headline-only run per back-end / family, so every table describes ONE workload
prof() {  # name, bench args...
  local name=$1; shift
  # PROFS="fp32 fp32_2lanes": only the named tables
  if [ -n "$PROFS" ]; then case " $PROFS " in *" $name "*) ;; *) return 0;; esac; fi
  rm -rf "$OUT/prof_$name"
  # --lanes 1: every launch behind the previous one, so that a kernel's begin -> end in the table is its own duration
  local base="--headline-only --steps 20 --windows 1 --lanes 1 --sustain-s 0"
  [ "$name" = plda ] && base=""
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o p -- python "$REPO/bench.py" $base "$@" > "$OUT/prof_$name.log" 2>&1
  python "$REPO/tools/rocprof_summary.py" "$(ls $OUT/prof_$name/*.db 2>/dev/null | head -1)" > "$OUT/${TAG}_kernel_stats_$name.md" 2>/dev/null
  head -4 "$OUT/${TAG}_kernel_stats_$name.md" | cut -c1-150
  rm -rf "$OUT/prof_$name"            # the .db files are tens of MB: gpurun_out/ only travels back under 64 MiB
}
if want prof; then
prof fp32 --precision fp32
prof fp32_2lanes --precision fp32 --lanes 2      # the headline mode: two batches in flight (+ its one-lane windows)
prof plda --plda-only --no-cpu-baseline
prof f16 --precision f16
prof f16x3 --precision f16x3
prof ResNet34_f16 --model ResNet34 --precision f16 --steps 5
prof ResNet34_fp32 --model ResNet34 --precision fp32 --steps 5
prof ResNet221_f16 --model ResNet221 --precision f16 --steps 5
prof ResNet221_fp32 --model ResNet221 --precision fp32 --steps 4
prof CAMPPlus_f16 --model CAMPPlus --precision f16 --steps 5
prof CAMPPlus_fp32 --model CAMPPlus --precision fp32 --steps 5
fi
# PMC passes (counters in their own runs, kernel-trace only): HBM traffic + MFMA-busy of the dominant class
pmc() {  # prec, needle, [model]
  local prec=$1 needle=$2 model=${3:-ECAPA_TDNN_GLOB_c512} i=0 tag=""
  [ "$model" != "ECAPA_TDNN_GLOB_c512" ] && tag="_$model"
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf "$OUT/pmc_${prec}_$i"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${prec}_$i" -- python "$REPO/bench.py" --model $model --precision $prec --steps 3 --warmup 1 --windows 1 --headline-only --lanes 1 > /dev/null 2>&1
  done
  python "$REPO/tools/pmc_traffic.py" $prec "$needle" "$OUT/pmc_${prec}_1" "$OUT/pmc_${prec}_2" "$OUT/pmc_${prec}_3" > "$OUT/${TAG}_pmc_dominant_kernel_$prec$tag.json"
  grep -E "traffic_bytes_per_launch|mfma_busy|\"traffic_bytes\"" "$OUT/${TAG}_pmc_dominant_kernel_$prec$tag.json"
  rm -rf "$OUT/pmc_${prec}_1" "$OUT/pmc_${prec}_2" "$OUT/pmc_${prec}_3"
}
if want pmc; then
pmc fp32 "gemm_f32_stream_kernel|astp_fused_kernel|conv_gemm_dual_kernel|conv_gemm_kernel<128, 128, 2, 2|conv_gemm_kernel<64, 64, 2, 2"
pmc f16 "gemm_f16_dma_kernel<128, 128|gemm_f16_p8_kernel|gemm_f16_dma_kernel<64, 64"
# the 2-D families: whole-forward HBM bytes tell whether their MFMA fraction is the binding limit at all
pmc f16 "gemm_f16_dma_kernel|gemm_f16_p8_kernel|conv3x3_direct_f16_kernel" ResNet221
pmc fp32 "gemm_f32_stream_kernel|conv_gemm_dual_kernel|conv_gemm_kernel" ResNet34
pmc fp32 "cam_dense_block_kernel|cam_dense_layer_kernel|gemm_f32_stream_kernel|conv_gemm_dual_kernel|conv_gemm_kernel|conv3x3_direct_f32_kernel" CAMPPlus
fi
cd "$REPO"
ls "$OUT" | grep "^${TAG}_" | tr '\n' ' '
