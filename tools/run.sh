#!/bin/bash
# tools/run.sh <name>: the command sequences of the GPU calls, one table instead of one launcher per call
# (gpurun -- 'tools/run.sh r05a').  Each entry names the gpurun_out/ files it writes; the ones quoted in DESIGN.md are
# copied to profiles/ and committed.  Round 4's one-shot launchers (tools/r04*.sh) are in the git history (commit a541f36).
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
NAME=${1:?usage: tools/run.sh <name>}
shift
case "$NAME" in
  r05a)   # the parity closures of VERDICT r04 next #1 (configs[4] length, Stage II / DINOv2 / AR windows at shipped architectures, LN fold stress, fp8 x f16)
    timeout 1500 python -m pytest -q -m gpu -rA tests/test_long64_gpu.py tests/test_autoencoder.py tests/test_image_encoder.py \
      tests/test_ln_fold_gpu.py "tests/test_denoiser_gpu.py::test_autoregressive_windows_configs2_at_the_headline_architecture" \
      "tests/test_baseline_arch_gpu.py::test_float16_mode_with_fp8_attention" \
      "tests/test_attention_fp8.py::test_fp8_fast_on_a_short_stream_runs_the_exact_form" 2>&1 | grep -v "^$" > gpurun_out/r05a_parity_tests.txt
    grep -E "passed|failed|error|rel-L2|relative error|LN fold stress|FAILED|ERROR|Error" gpurun_out/r05a_parity_tests.txt | cut -c1-260 | tail -90
    ;;
  *) echo "unknown entry $NAME"; exit 2 ;;
esac
