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
  r05b)   # the rest of the parity closures (r05a ran out of its clock: 25 min, oracle legs on 128+ host threads), the GELU table, the e4m3 forms on the
          # peaky fixtures, the N > 1 bench (A/B + fallback + fingerprint self-check), then GEMM / attention timings and the energy table
    PT="python -m pytest -q -m gpu -v --timeout=420 --durations=15"
    timeout 1100 $PT "tests/test_long64_gpu.py::test_attention_long64_key_coverage" tests/test_autoencoder.py \
      "tests/test_image_encoder.py::test_hip_encoder_vitl_against_transformers" \
      "tests/test_denoiser_gpu.py::test_autoregressive_windows_configs2_at_the_headline_architecture" \
      "tests/test_baseline_arch_gpu.py::test_float16_mode_with_fp8_attention" "tests/test_baseline_arch_gpu.py::test_peaky_attention_fp8_forms" \
      tests/test_attention_fp8.py "tests/test_kernels_gpu.py::test_gelu_table_is_bit_identical" "tests/test_kernels_gpu.py::test_gemm" \
      "tests/test_kernels_gpu.py::test_gemm256_pingpong_main_loop" 2>&1 | grep -v "^$" | grep -vE "PASSED|^tests/.*(SKIPPED)" > gpurun_out/r05b_tests.txt
    tail -45 gpurun_out/r05b_tests.txt | cut -c1-250
    timeout 700 $PT -s tests/test_multi_gpu.py -k "bench or phase_loop or copy_engine_exchange_across" 2>&1 | grep -v "^$" | grep -vE "PASSED" | tail -25 | cut -c1-300 | tee gpurun_out/r05b_mgpu.txt
    if [ "$1" != "tests-only" ]; then
    python tools/kernel_bench.py --only gemm --product-only --blas --reps 20 2>&1 | grep -E "^gemm" | tee gpurun_out/r05b_gemm.txt
    python tools/kernel_bench.py --only attn --product-only --fp8 --reps 5 2>&1 | tail -16 | tee gpurun_out/r05b_attn.txt
    fi
    ACTIONMESH_AMD_LIB=build/variants/libam_abl.so python tools/limiter_probe.py --energy-table --seconds 1.5 --out gpurun_out/r05b_energy_table.json 2>&1 | tail -20 | tee gpurun_out/r05b_energy_table.txt
    for t in $([ "$1" != "tests-only" ] && echo 1 0); do
      ACTIONMESH_AMD_GELU_TABLE=$t timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r05b_bench_gelutab$t.json
      python -c "
import json; d=json.load(open('gpurun_out/r05b_bench_gelutab$t.json'))
print('gelu table $t:', {k: d[k] for k in ('value','ms_per_step')}, d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline'].get('energy_j'), d['roofline'].get('pj_per_flop'), d['latents_fingerprint'])"
    done
    ;;
  r05c)   # restated tests (AR configs[2] vs the reduced-precision yardstick, the e4m3 forms on the peaky fixtures), the N > 1 bench at the HEADLINE shape
          # and the driver's step counts on one device (does the fingerprint self-check hold at 25 steps?), the energy table with the
          # ablation build, the "next" rows re-measured on today's kernels in both 16-bit types
    PT="python -m pytest -q -m gpu -rA --timeout=420 --durations=10"
    timeout 600 $PT "tests/test_denoiser_gpu.py::test_autoregressive_windows_configs2_at_the_headline_architecture" \
      "tests/test_baseline_arch_gpu.py::test_peaky_attention_fp8_forms" 2>&1 | grep -v "^$" > gpurun_out/r05c_tests.txt
    grep -E "passed|failed|rel-L2|FAILED|Error" gpurun_out/r05c_tests.txt | cut -c1-300 | tail -12
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 \
      --same-device --steps 20 --warmup 5 --no-roofline 2>gpurun_out/r05c_bench4.err | grep "^{" > gpurun_out/r05c_bench_same_device_4.json
    python -c "
import json; d=json.load(open('gpurun_out/r05c_bench_same_device_4.json'))
print('same-device x4 headline:', d['ms_per_step'], d['exchange_ab'], d['fingerprint_check'])" || tail -5 gpurun_out/r05c_bench4.err
    ACTIONMESH_AMD_LIB=build/variants/libam_abl.so python tools/limiter_probe.py --energy-table --seconds 1.5 --out gpurun_out/r05c_energy_table.json 2>&1 | grep -v "^{" | tail -22 | tee gpurun_out/r05c_energy_table.txt
    for dt in bfloat16 float16; do
      python tools/stage2_bench.py --dtype $dt 2>/dev/null | tail -1 | tee gpurun_out/r05c_stage2_$dt.json | cut -c1-330
      python tools/encoder_bench.py --dtype $dt 2>/dev/null | tail -1 | tee gpurun_out/r05c_encoder_$dt.json | cut -c1-330
    done
    python tools/e2e_synthetic.py 2>/dev/null | tail -1 | tee gpurun_out/r05c_e2e_synthetic.json | cut -c1-420
    ;;
  r05final)   # end-of-round evidence for the FINAL sources: the whole GPU suite (per-test output + durations kept), smoke, the contract bench line,
              # the profile set named by the sources sha, configs[4] on one device (fp8 / fp8_fast) behind its parity tests, the fp8 headline lines
    TAG=r05
    timeout 2400 python -m pytest tests -q -m gpu -rA --timeout=600 --durations=40 2>&1 | grep -v "^$" > gpurun_out/${TAG}_gputest_full.txt
    grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest_full.txt | tail -15 | tee gpurun_out/${TAG}_gputest.txt
    grep -A42 "slowest 40 durations" gpurun_out/${TAG}_gputest_full.txt | head -44 >> gpurun_out/${TAG}_gputest.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee -a gpurun_out/${TAG}_gputest.txt
    timeout 900 python bench.py --steps 10 --warmup 2 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_headline.json
    python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline.json'))
r=d['roofline']; c=d['cpu_baseline']
print({k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_bf16_peak')}, r['launch_ms'], r['frac'], r['traffic'], r.get('energy_j'), r.get('pj_per_flop'), r.get('effective_clock_ghz'), r.get('pipe_busy'), d.get('nominal',{}).get('ms_per_step'), d['with_exact_shortcuts']['ms_per_step'])
print('cpu_baseline', c['value'], c['cores'], c['derivation'], c['scaled_reference'], c['fit']['value'], c['fit']['max_relative_residual'])"
    tools/gpu_profile.sh $TAG 2>&1 | tail -8
    for dt in fp8 fp8_fast; do
      timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_headline_$dt.json
      timeout 600 python bench.py --shape long64 --dtype $dt --steps 1 --warmup 1 --no-cpu-baseline --no-nominal --no-roofline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_long64_$dt.json
    done
    timeout 600 python bench.py --shape long64 --dtype fp8 --emulate-world 8 --steps 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_emulate8_long64_fp8.json
    python -c "
import json
for f in ('bench_headline_fp8', 'bench_headline_fp8_fast', 'bench_long64_fp8', 'bench_long64_fp8_fast', 'emulate8_long64_fp8'):
    d = json.load(open('gpurun_out/${TAG}_' + f + '.json')); print(f, {k: d[k] for k in d if k in ('ms_per_step', 'value', 'dtype', 'step_frac_of_dtype_peak', 'rank0_ms_per_step', 'modelled_link_ms_per_layer', 'step_tflops_per_gpu')}, (d.get('roofline') or {}).get('launch_ms'), (d.get('roofline') or {}).get('frac'))"
    ;;
  r05d)   # after the final run: the tests whose tolerances were tightened to the measured values, and the one-device PROJECTION of 2 / 4 / 8 ranks
          # (rank 0's share + modelled link time) on this round's kernels
    timeout 600 python -m pytest -q -m gpu -rA --timeout=420 "tests/test_autoencoder.py::test_hip_autoencoder_at_the_shipped_architecture" \
      "tests/test_long64_gpu.py::test_one_layer_model_at_long64" 2>&1 | grep -E "passed|failed|rel-L2|FAILED" | cut -c1-300 | tee gpurun_out/r05d_tests.txt
    for P in 2 4 8; do
      timeout 300 python bench.py --emulate-world $P --steps 3 2>/dev/null | tail -1 > gpurun_out/r05d_emulate_world_$P.json
      python -c "
import json; d=json.load(open('gpurun_out/r05d_emulate_world_$P.json')); print('P=$P', {k: d[k] for k in d if k in ('rank0_ms_per_step','modelled_link_ms_per_layer','host_enqueue_ms_per_step','local_pass_ms_per_layer')})"
    done
    ;;
  r05e)   # robustness of the hardened copy-engine exchange (fine-grained flags, ring-owned events) between real processes, repeated; the pure
          # frame-sharding partition at the headline shape on one device; one driver-style bench line (the driver's own step counts)
    tools/peer_stress.sh 10 2 --loop both --forwards 4 2>&1 | tail -3 | tee gpurun_out/r05e_peer_stress.txt
    tools/peer_stress.sh 8 4 2>&1 | tail -3 | tee -a gpurun_out/r05e_peer_stress.txt
    tools/peer_stress.sh 4 4 --dtype fp8 2>&1 | tail -3 | tee -a gpurun_out/r05e_peer_stress.txt
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 4 \
      --same-device --cfg-parallel 0 --steps 20 --warmup 5 --no-roofline 2>gpurun_out/r05e_bench4.err | grep "^{" > gpurun_out/r05e_bench_same_device_4_frames_only.json
    python -c "
import json; d=json.load(open('gpurun_out/r05e_bench_same_device_4_frames_only.json'))
print('same-device x4 pure frame sharding:', d['config']['parallelism'], d['ms_per_step'], d['exchange_ab'], d['fingerprint_check']['legs'], d['fingerprint_ok'])" || tail -5 gpurun_out/r05e_bench4.err
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05e_bench_driver_style.json
    python -c "
import json; d=json.load(open('gpurun_out/r05e_bench_driver_style.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step','steps','warmup')}, r['launch_ms'], r['frac'], r['traffic'], r['traffic_source'], r.get('energy_j'), d['nominal']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['derivation'][:20])"
    ;;
  r05f)   # the packed row sums of the 4x64 attention kernel (AM_A64_PKSUM): same-box A/B against the scalar adds, bit identity, the attention tests
    for v in nopk pk; do ACTIONMESH_AMD_LIB=build/variants/libam_$v.so python tools/diag/attn_bits.py 2>/dev/null > gpurun_out/r05f_bits_$v.txt; done
    if cmp -s gpurun_out/r05f_bits_nopk.txt gpurun_out/r05f_bits_pk.txt; then echo "BIT-IDENTICAL: packed vs scalar row sums"; else echo "DIFFERENT BITS"; diff gpurun_out/r05f_bits_nopk.txt gpurun_out/r05f_bits_pk.txt; fi
    cat gpurun_out/r05f_bits_pk.txt
    tools/ab_attn64.sh "nopk pk" "" "" > /dev/null 2>&1; grep -E "===|variant=  8" gpurun_out/ab_attn64.txt | tee gpurun_out/r05f_ab_pksum.txt
    timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_f16_gpu.py -q -m gpu -k "attention" --timeout=300 2>&1 | tail -3 | tee gpurun_out/r05f_tests.txt
    for t in nopk pk; do
      ACTIONMESH_AMD_LIB=build/variants/libam_$t.so timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r05f_bench_$t.json
      python -c "
import json; d=json.load(open('gpurun_out/r05f_bench_$t.json')); r=d['roofline']
print('$t:', {k: d[k] for k in ('value','ms_per_step')}, r['launch_ms'], r['frac'], r.get('energy_j'), r.get('effective_clock_ghz'), r.get('pipe_busy'), d['latents_fingerprint']['sample'][:3])"
    done
    ;;
  r06a)   # round 6, first contact: the self-launching bench + pre-flight on one device, then the GEMM tables the model
          # really launches (fused epilogues) at both architectures, and a baseline bench line of the unchanged kernels
    PT="python -m pytest -q -m gpu -v --timeout=600 --durations=10"
    timeout 1200 $PT tests/test_multi_gpu.py -k "bench" 2>&1 | grep -v "^$" | grep -vE "PASSED" | tail -30 | cut -c1-400 | tee gpurun_out/r06a_mgpu.txt
    for sh in headline nominal; do
      python tools/kernel_bench.py --shape $sh --only gemm,fused --product-only --blas --reps 20 2>&1 | grep -E "^gemm|^qkv|^cross" | tee gpurun_out/r06a_gemm_$sh.txt
    done
    timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06a_bench.json
    python -c "
import json; d=json.load(open('gpurun_out/r06a_bench.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step')}, r['launch_ms'], r['frac'], d['nominal']['ms_per_step'])"
    ;;
  r06b)   # the hoisted fused-QKV epilogue + norm_out fold: kernel / fold / model parity tests, the in-model QKV launch with its ablations, a bench line
    PT="python -m pytest -q -m gpu --timeout=600 --durations=8"
    timeout 1500 $PT tests/test_kernels_gpu.py tests/test_ln_fold_gpu.py tests/test_denoiser_gpu.py tests/test_f16_gpu.py -x 2>&1 | grep -v "^$" | tail -25 | cut -c1-300 | tee gpurun_out/r06b_tests.txt
    for sh in headline nominal; do
      python tools/kernel_bench.py --shape $sh --only fused --reps 20 2>&1 | grep -E "qkv|cross" | tee gpurun_out/r06b_fused_$sh.txt
    done
    timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06b_bench.json
    python -c "
import json; d=json.load(open('gpurun_out/r06b_bench.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step')}, r['launch_ms'], r['frac'], d['nominal']['ms_per_step'], d['latents_fingerprint'])"
    ;;
  r06c)   # fp32 residual streams (Stage II, DINOv2), the batched QK epilogue, norm_out fold: tests + timings
    PT="python -m pytest -q -m gpu --timeout=600 --durations=8 -rA"
    timeout 1800 $PT tests/test_kernels_gpu.py tests/test_ln_fold_gpu.py tests/test_denoiser_gpu.py tests/test_f16_gpu.py tests/test_autoencoder.py tests/test_image_encoder.py \
      2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|ERROR|Stage II at|HIP DINOv2|rel-L2|Error" | cut -c1-330 | tee gpurun_out/r06c_tests.txt
    python tools/kernel_bench.py --shape headline --only fused --reps 20 2>&1 | grep -E "qkv|cross" | tee gpurun_out/r06c_fused_headline.txt
    for dt in bfloat16 float16; do
      for rf in 1 0; do
        ACTIONMESH_AMD_RESIDUAL_FP32=$rf python tools/stage2_bench.py --dtype $dt 2>/dev/null | tail -1 | tee gpurun_out/r06c_stage2_${dt}_res$rf.json | cut -c1-330
        ACTIONMESH_AMD_RESIDUAL_FP32=$rf python tools/encoder_bench.py --dtype $dt 2>/dev/null | tail -1 | tee gpurun_out/r06c_encoder_${dt}_res$rf.json | cut -c1-330
      done
    done
    timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06c_bench.json
    python -c "
import json; d=json.load(open('gpurun_out/r06c_bench.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step')}, r['launch_ms'], r['frac'], d['nominal']['ms_per_step'], d['latents_fingerprint'])"
    ;;
  r06d)   # probes: joules per flop of the two bf16 MFMA shapes; the attention kernel at the nominal head count; a kernel
          # trace of the NOMINAL step (next #2 v); LDS bank conflicts of the ping-pong GEMM per launch shape (next #2 iv)
    export TMPDIR=/tmp
    python tools/ubench/mfma_energy.py --seconds 3 --out gpurun_out/r06d_mfma_energy.json 2>&1 | tail -22 | tee gpurun_out/r06d_mfma_energy.txt
    python tools/kernel_bench.py --shape nominal --only attn,xattn --product-only --reps 5 2>&1 | tail -6 | tee gpurun_out/r06d_attn_nominal.txt
    python tools/kernel_bench.py --shape headline --only attn,xattn --product-only --reps 5 2>&1 | tail -6 | tee gpurun_out/r06d_attn_headline.txt
    OUT=$PWD/gpurun_out/prof_r06d_nominal; rm -rf $OUT; mkdir -p $OUT
    timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --shape nominal --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
    python tools/summarize_prof.py $OUT gpurun_out/r06d_nominal 2>&1 | tail -3
    head -14 gpurun_out/r06d_nominal_by_launch_shape.csv | cut -c1-200
    for nm in qkv "ff1+gelu" "ff2+res" "attn-out+res"; do
      P=$PWD/gpurun_out/prof_r06d_lds_$(echo $nm | tr -d '+()'); rm -rf $P; mkdir -p $P
      timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES -d $P/pmc_lds -o pmc -- python tools/kernel_bench.py --only gemm --product-only --reps 1 --gemm-names "$nm" > $P/log.txt 2>&1
      python tools/summarize_prof.py $P gpurun_out/r06d_lds_$(echo $nm | tr -d '+()') 2>&1 | tail -1
      grep -i gemm256 gpurun_out/r06d_lds_$(echo $nm | tr -d '+()')_pmc.csv | cut -c1-260
    done
    ;;
  r06e)   # does the cheaper MFMA shape (r06d: 16x16x32 moves 8 % fewer joules per flop than 32x32x16 in a bare stream) pay INSIDE the 4x64 attention
          # kernel?  TIMING probe only (tools/gen_attn64_asm.py AM_A64_MFMA16_TIMING: wrong results by construction): P.V phase as 2 x 16x16x32
    V=$PWD/build/variants
    for round in 1 2; do for v in base pv16; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee gpurun_out/r06e_pv16_ab.txt
    for v in base pv16; do
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r06e_bench_$v.json
      python -c "
import json; d=json.load(open('gpurun_out/r06e_bench_$v.json')); r=d['roofline']
print('$v:', d['ms_per_step'], r['launch_ms'], r['frac'], 'J', r.get('energy_j'), 'GHz', r.get('effective_clock_ghz'), 'busy', r.get('pipe_busy'), 'W', (r.get('clock_telemetry') or {}).get('average_power_w'), 'zero-operand ms', (r['samples'].get('zero operands (diagnostic: same launch, nothing toggles - the schedule\'s rate at the full clock)') or {}).get('launch_ms'))" | tee -a gpurun_out/r06e_pv16_ab.txt
    done
    ;;
  r06f)   # operand-order probe of the 4x64 attention kernel (AM_A64_SNAKE: every MFMA step changes ONE operand instead of one-then-both): bits, time, joules
    V=$PWD/build/variants
    for v in base snake; do ACTIONMESH_AMD_LIB=$V/libam_$v.so python tools/diag/attn_bits.py 2>/dev/null > gpurun_out/r06f_bits_$v.txt; done
    if cmp -s gpurun_out/r06f_bits_base.txt gpurun_out/r06f_bits_snake.txt; then echo "BIT-IDENTICAL: snake vs plain MFMA order"; else echo "DIFFERENT BITS"; fi | tee gpurun_out/r06f_snake_ab.txt
    for round in 1 2 3; do for v in base snake; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee -a gpurun_out/r06f_snake_ab.txt
    for v in base snake base snake; do
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r06f_bench_$v.json
      python -c "
import json; d=json.load(open('gpurun_out/r06f_bench_$v.json')); r=d['roofline']
print('$v:', d['ms_per_step'], r['launch_ms'], r['frac'], 'J', r.get('energy_j'), 'GHz', r.get('effective_clock_ghz'), 'busy', r.get('pipe_busy'), 'W', (r.get('clock_telemetry') or {}).get('average_power_w'))" | tee -a gpurun_out/r06f_snake_ab.txt
    done
    ;;
  r06final)   # end-of-round evidence for the FINAL sources of round 6: the whole GPU suite, smoke, the driver's own bench command, the profile set named by
              # the sources sha, the in-model GEMM table, the fp8 lines, configs[1] / [3] plumbing records, configs[4] on one device, N > 1 dry runs
    TAG=r06
    timeout 2700 python -m pytest tests -q -m gpu -rA --timeout=600 --durations=40 2>&1 | grep -v "^$" > gpurun_out/${TAG}_gputest_full.txt
    grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest_full.txt | tail -15 | tee gpurun_out/${TAG}_gputest.txt
    grep -A42 "slowest 40 durations" gpurun_out/${TAG}_gputest_full.txt | head -44 >> gpurun_out/${TAG}_gputest.txt
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee -a gpurun_out/${TAG}_gputest.txt
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_headline.json
    python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline.json'))
r=d['roofline']; c=d['cpu_baseline']
print({k: d[k] for k in ('value','ms_per_step','dtype','step_frac_of_bf16_peak')}, r['launch_ms'], r['frac'], r['traffic'], r.get('energy_j'), r.get('pj_per_flop'), r.get('effective_clock_ghz'), r.get('pipe_busy'), d.get('nominal',{}).get('ms_per_step'), d['with_exact_shortcuts']['ms_per_step'])
print('non-attention ms per step', round(d['ms_per_step'] - 21 * r['launch_ms'], 2))
print('cpu_baseline', c['value'], c['cores'], c['derivation'], c['scaled_reference'], c['fit']['value'], c['fit']['max_relative_residual'])"
    tools/gpu_profile.sh $TAG 2>&1 | tail -8
    for sh in headline nominal; do
      python tools/kernel_bench.py --shape $sh --only gemm,fused --product-only --blas --reps 20 2>&1 | grep -E "^gemm|qkv|^cross" > gpurun_out/${TAG}_gemm_$sh.txt
    done
    grep -E "in-model|ablation|256sq-pingpong:|hipBLASLt" gpurun_out/${TAG}_gemm_headline.txt | cut -c1-200
    for dt in fp8 fp8_fast; do
      timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_headline_$dt.json
      timeout 600 python bench.py --shape long64 --dtype $dt --steps 1 --warmup 1 --no-cpu-baseline --no-nominal --no-roofline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_long64_$dt.json
    done
    python -c "
import json
for f in ('bench_headline_fp8', 'bench_headline_fp8_fast', 'bench_long64_fp8', 'bench_long64_fp8_fast'):
    d = json.load(open('gpurun_out/${TAG}_' + f + '.json')); print(f, {k: d[k] for k in d if k in ('ms_per_step', 'value', 'dtype', 'step_frac_of_dtype_peak', 'step_tflops_per_gpu')}, (d.get('roofline') or {}).get('launch_ms'), (d.get('roofline') or {}).get('frac'))"
    for c in 1 3; do
      python tools/e2e_synthetic.py --config $c 2>/dev/null | tail -1 | tee gpurun_out/${TAG}_config$c.json | cut -c1-600
    done
    python tools/e2e_synthetic.py 2>/dev/null | tail -1 | tee gpurun_out/${TAG}_e2e_synthetic.json | cut -c1-420
    for P in 2 4 8; do
      timeout 300 python bench.py --emulate-world $P --steps 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_emulate_world_$P.json
    done
    timeout 600 python bench.py --shape long64 --dtype fp8 --emulate-world 8 --steps 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_emulate8_long64_fp8.json
    python -c "
import json
for f in ('emulate_world_2', 'emulate_world_4', 'emulate_world_8', 'emulate8_long64_fp8'):
    d = json.load(open('gpurun_out/${TAG}_' + f + '.json')); print(f, {k: d[k] for k in d if k in ('rank0_ms_per_step', 'modelled_link_ms_per_layer')})"
    timeout 900 python bench.py --gpus 4 --same-device --steps 20 --warmup 5 --no-roofline 2>gpurun_out/${TAG}_bench4.err | grep "^{" > gpurun_out/${TAG}_bench_same_device_x4_headline.json
    python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_same_device_x4_headline.json'))
print('self-launched same-device x4 headline:', d['ms_per_step'], d['exchange_ab'], d['fingerprint_check'], d['launcher']['attempts'])" || tail -5 gpurun_out/${TAG}_bench4.err
    ;;
  r06g)   # after the final run: the pre-flight tweak (children leave together) under the multi-rank bench tests, and the driver's own bench command on
          # another box - now with roofline.traffic resolved from the committed record of this sources sha
    timeout 900 python -m pytest -q -m gpu --timeout=600 tests/test_multi_gpu.py -k "bench" 2>&1 | tail -3 | tee gpurun_out/r06g_mgpu.txt
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06g_bench_driver_style.json
    python -c "
import json; d=json.load(open('gpurun_out/r06g_bench_driver_style.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step','steps','warmup','step_frac_of_bf16_peak')}, r['launch_ms'], r['frac'], r['traffic'], r['traffic_source'], r.get('energy_j'), d['nominal']['ms_per_step'], d['with_exact_shortcuts']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('non-attention ms per step', round(d['ms_per_step'] - 21 * r['launch_ms'], 2))"
    ;;
  r06h)   # the resident-key-stream cross-attention kernel: tests, the key-count sweep with and without it, same-box step A/B
    PT="python -m pytest -q -m gpu --timeout=600"
    timeout 1500 $PT tests/test_kernels_gpu.py -k "attention" 2>&1 | tail -4 | tee gpurun_out/r06h_tests.txt
    timeout 1500 $PT tests/test_baseline_arch_gpu.py tests/test_denoiser_gpu.py tests/test_f16_gpu.py -x 2>&1 | tail -4 | tee -a gpurun_out/r06h_tests.txt
    for r in 0 1; do
      echo "=== ACTIONMESH_AMD_XATTN_RESIDENT=$r"
      ACTIONMESH_AMD_XATTN_RESIDENT=$r python tools/kernel_bench.py --shape headline --only xsweep --reps 10 2>&1 | grep "sweep"
      ACTIONMESH_AMD_XATTN_RESIDENT=$r python tools/kernel_bench.py --shape nominal --only xsweep --reps 10 2>&1 | grep "S= 257"
    done | tee gpurun_out/r06h_xsweep.txt
    for r in 0 1 0 1; do
      ACTIONMESH_AMD_XATTN_RESIDENT=$r timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06h_bench_res$r.json
      python -c "
import json; d=json.load(open('gpurun_out/r06h_bench_res$r.json'))
print('resident=$r:', d['ms_per_step'], d['nominal']['ms_per_step'], d['with_exact_shortcuts']['ms_per_step'], d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06h_xsweep.txt
    done
    ;;
  r06k)   # resident cross-attention, third form (output staged through LDS: 64-byte store segments; compact tail V^T): tests + sweep + step A/B
    PT="python -m pytest -q -m gpu --timeout=600"
    timeout 900 $PT tests/test_kernels_gpu.py -k "resident or test_attention" 2>&1 | tail -3 | tee gpurun_out/r06k_tests.txt
    timeout 1200 $PT tests/test_baseline_arch_gpu.py -k "full or forward" tests/test_denoiser_gpu.py -k "full_size or forward or sharded or world" tests/test_f16_gpu.py 2>&1 | tail -3 | tee -a gpurun_out/r06k_tests.txt
    for r in 0 1; do
      echo "=== ACTIONMESH_AMD_XATTN_RESIDENT=$r"
      ACTIONMESH_AMD_XATTN_RESIDENT=$r python tools/kernel_bench.py --shape headline --only xsweep --reps 10 2>&1 | grep "sweep"
      ACTIONMESH_AMD_XATTN_RESIDENT=$r python tools/kernel_bench.py --shape nominal --only xsweep --reps 10 2>&1 | grep "S= 257"
    done | tee gpurun_out/r06k_xsweep.txt
    for r in 0 1 0 1; do
      ACTIONMESH_AMD_XATTN_RESIDENT=$r timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06k_bench_res$r.json
      python -c "
import json; d=json.load(open('gpurun_out/r06k_bench_res$r.json'))
print('resident=$r:', d['ms_per_step'], d['nominal']['ms_per_step'], d['with_exact_shortcuts']['ms_per_step'], d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06k_xsweep.txt
    done
    ;;
  r06i)   # resident cross-attention, second form (Q pipeline re-ordered, query blocks cut over workgroups when pairs are few): tests + sweep + step A/B
    PT="python -m pytest -q -m gpu --timeout=600"
    timeout 900 $PT tests/test_kernels_gpu.py -k "resident or test_attention" 2>&1 | tail -3 | tee gpurun_out/r06i_tests.txt
    timeout 1200 $PT tests/test_baseline_arch_gpu.py -k "full or forward" tests/test_denoiser_gpu.py -k "full_size or forward or sharded or world" 2>&1 | tail -3 | tee -a gpurun_out/r06i_tests.txt
    for r in 0 1; do
      echo "=== ACTIONMESH_AMD_XATTN_RESIDENT=$r"
      ACTIONMESH_AMD_XATTN_RESIDENT=$r python tools/kernel_bench.py --shape headline --only xsweep --reps 10 2>&1 | grep "sweep"
      ACTIONMESH_AMD_XATTN_RESIDENT=$r python tools/kernel_bench.py --shape nominal --only xsweep --reps 10 2>&1 | grep "S= 257"
    done | tee gpurun_out/r06i_xsweep.txt
    for r in 0 1 0 1; do
      ACTIONMESH_AMD_XATTN_RESIDENT=$r timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06i_bench_res$r.json
      python -c "
import json; d=json.load(open('gpurun_out/r06i_bench_res$r.json'))
print('resident=$r:', d['ms_per_step'], d['nominal']['ms_per_step'], d['with_exact_shortcuts']['ms_per_step'], d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06i_xsweep.txt
    done
    ;;
  r06j)   # where the exact e4m3 attention's tile time goes, on this round's build ("or an ablation table that shows the floor"):
          # the product (free-running) kernel's timing ablations + the 8-wave form's, same box
    ACTIONMESH_AMD_LIB=$PWD/build/variants/libam_fp8prof.so python tools/kernel_bench.py --only attn --product-only --fp8 --ablate-fp8 --reps 3 2>&1 | grep -E "fp8|self-attn" | tee gpurun_out/r06j_fp8_ablations.txt
    ;;
  r06m)   # row sums over the ROUNDED probabilities as v_dot2c_f32_bf16 (AM_A64_DOTSUM: 64 softmax steps per block instead of 80): distance to fp64
          # of both builds, interleaved launch timings, joules, the attention tests on the candidate, same-box step A/B
    V=$PWD/build/variants
    for v in sum dot; do echo "=== $v"; ACTIONMESH_AMD_LIB=$V/libam_$v.so python tools/diag/attn_accuracy.py 2>/dev/null; done | tee gpurun_out/r06m_accuracy.txt
    for round in 1 2 3; do for v in sum dot; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee gpurun_out/r06m_dotsum_ab.txt
    ACTIONMESH_AMD_LIB=$V/libam_dot.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" --timeout=300 2>&1 | tail -3 | tee gpurun_out/r06m_tests.txt
    for v in sum dot sum dot; do
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r06m_bench_$v.json
      python -c "
import json; d=json.load(open('gpurun_out/r06m_bench_$v.json')); r=d['roofline']
print('$v:', d['ms_per_step'], r['launch_ms'], r['frac'], 'J', r.get('energy_j'), 'GHz', r.get('effective_clock_ghz'), 'busy', r.get('pipe_busy'), 'W', (r.get('clock_telemetry') or {}).get('average_power_w'), d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06m_dotsum_ab.txt
    done
    ;;
  r06n)   # this round's build under the firmware telemetry: the limiter probe (idle / N(0,1) / zero-operand legs: clock, power, PPT residency) and the
          # energy table of the ablation build (VERDICT r05 next #5 names profiles/r06_energy_table.*); the MFMA-shape rows are r06d_mfma_energy.*
    python tools/limiter_probe.py --seconds 2 --out gpurun_out/r06_limiter_probe.json 2>&1 | grep -v "^{" | tail -14 | tee gpurun_out/r06_limiter_probe.txt
    ACTIONMESH_AMD_LIB=build/variants/libam_abl.so python tools/limiter_probe.py --energy-table --seconds 1.5 --out gpurun_out/r06_energy_table.json 2>&1 | grep -v "^{" | tail -24 | tee gpurun_out/r06_energy_table.txt
    ;;
  r06p)   # where a phase's four LDS-DMA pieces sit (AM_A64_DMAPOS: 0 = first four even gaps, 1 = one per k-step quarter, 2 = P.V gaps 4-7, 3 = two pairs):
          # bit identity, interleaved launch timings, step A/B on one box
    V=$PWD/build/variants
    for v in dp0 dp1 dp2 dp3; do ACTIONMESH_AMD_LIB=$V/libam_$v.so python tools/diag/attn_bits.py 2>/dev/null > gpurun_out/r06p_bits_$v.txt; done
    for v in dp1 dp2 dp3; do if cmp -s gpurun_out/r06p_bits_dp0.txt gpurun_out/r06p_bits_$v.txt; then echo "BIT-IDENTICAL: $v vs dp0"; else echo "DIFFERENT BITS: $v"; fi; done | tee gpurun_out/r06p_dmapos_ab.txt
    for round in 1 2 3; do for v in dp0 dp1 dp2 dp3; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee -a gpurun_out/r06p_dmapos_ab.txt
    for v in dp0 dp1 dp2 dp3; do
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r06p_bench_$v.json
      python -c "
import json; d=json.load(open('gpurun_out/r06p_bench_$v.json')); r=d['roofline']
print('$v:', d['ms_per_step'], r['launch_ms'], r['frac'], 'J', r.get('energy_j'), 'GHz', r.get('effective_clock_ghz'), 'busy', r.get('pipe_busy'), 'W', (r.get('clock_telemetry') or {}).get('average_power_w'), d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06p_dmapos_ab.txt
    done
    ;;
  r06q)   # v_exp_legacy_f32: rate / price beside MFMAs / accuracy (tools/ubench/exp_legacy.hip), then the 4x64 attention kernel with it
          # (AM_A64_EXPLEGACY): distance to fp64, interleaved launch timings, step A/B
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/exp_legacy tools/ubench/exp_legacy.hip && /tmp/exp_legacy 2>&1 | tee gpurun_out/r06q_exp_legacy_ubench.txt
    V=$PWD/build/variants
    for v in el0 el1; do echo "=== $v"; ACTIONMESH_AMD_LIB=$V/libam_$v.so python tools/diag/attn_accuracy.py 2>/dev/null; done | tee gpurun_out/r06q_accuracy.txt
    for round in 1 2 3; do for v in el0 el1; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee gpurun_out/r06q_explegacy_ab.txt
    for v in el0 el1 el0 el1; do
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r06q_bench_$v.json
      python -c "
import json; d=json.load(open('gpurun_out/r06q_bench_$v.json')); r=d['roofline']
print('$v:', d['ms_per_step'], r['launch_ms'], r['frac'], 'J', r.get('energy_j'), 'GHz', r.get('effective_clock_ghz'), 'busy', r.get('pipe_busy'), 'W', (r.get('clock_telemetry') or {}).get('average_power_w'), d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06q_explegacy_ab.txt
    done
    ;;
  r06r)   # block 1's first M exponentials beside the P.V MFMAs of phase 1 (AM_A64_EARLYEX; AGPR-form MFMAs hide ~6 VALU slots, the VGPR-form QK^T ones ~2):
          # bit identity, interleaved launch timings, step A/B on one box.  usage: tools/run.sh r06r "ee0 ee24 ee16 ..."
    V=$PWD/build/variants
    VARS=${1:-"ee0 ee24 ee16 ee24c0"}
    FIRST=$(echo $VARS | cut -d" " -f1)
    for v in $VARS; do ACTIONMESH_AMD_LIB=$V/libam_$v.so python tools/diag/attn_bits.py 2>/dev/null > gpurun_out/r06r_bits_$v.txt; done
    for v in $VARS; do if cmp -s gpurun_out/r06r_bits_$FIRST.txt gpurun_out/r06r_bits_$v.txt; then echo "BIT-IDENTICAL: $v vs $FIRST"; else echo "DIFFERENT BITS: $v"; fi; done | tee gpurun_out/r06r_earlyex_ab.txt
    for round in 1 2 3; do for v in $VARS; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee -a gpurun_out/r06r_earlyex_ab.txt
    for v in $VARS; do
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-nominal 2>/dev/null | tail -1 > gpurun_out/r06r_bench_$v.json
      python -c "
import json; d=json.load(open('gpurun_out/r06r_bench_$v.json')); r=d['roofline']
print('$v:', d['ms_per_step'], r['launch_ms'], r['frac'], 'J', r.get('energy_j'), 'GHz', r.get('effective_clock_ghz'), 'busy', r.get('pipe_busy'), 'W', (r.get('clock_telemetry') or {}).get('average_power_w'), d['latents_fingerprint']['rms'])" | tee -a gpurun_out/r06r_earlyex_ab.txt
    done
    ;;
  r06s)   # cache-policy bits on the operand LDS-DMA loads (default / nt / sc1 / sc0 sc1 / sc1 nt): attention K / V^T streams (AM_A64_DMA_CPOL) and the
          # ping-pong GEMM's A / W stages (AM_G_AUX_A / _W), interleaved rounds on one box; bits must not change
    V=$PWD/build/variants
    for v in cp0 cp1 cp2 cp3 cp4; do ACTIONMESH_AMD_LIB=$V/libam_$v.so python tools/diag/attn_bits.py 2>/dev/null > gpurun_out/r06s_bits_$v.txt; done
    for v in cp1 cp2 cp3 cp4; do if cmp -s gpurun_out/r06s_bits_cp0.txt gpurun_out/r06s_bits_$v.txt; then echo "BIT-IDENTICAL: $v vs cp0"; else echo "DIFFERENT BITS: $v"; fi; done | tee gpurun_out/r06s_cpol_ab.txt
    for round in 1 2; do for v in cp0 cp1 cp2 cp3 cp4; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only attn --product-only --reps 6 2>&1 | grep "self-attn"
    done; done | tee -a gpurun_out/r06s_cpol_ab.txt
    for round in 1 2; do for v in g00 g22 g20 g02 gss gsn; do
      echo "=== round $round $v"
      ACTIONMESH_AMD_LIB=$V/libam_$v.so timeout 300 python tools/kernel_bench.py --only gemm --product-only --reps 20 2>&1 | grep "256sq-pingpong:"
    done; done | tee gpurun_out/r06s_gemm_cpol_ab.txt
    ;;
  r06t)   # after the final run, on another box: the driver's own bench command (roofline.traffic now resolved from the committed record of this sha) and a
          # kernel trace of the NOMINAL step on the final sources (r06d's predates the resident cross-attention kernel)
    export TMPDIR=/tmp
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06t_bench_driver_style.json
    python -c "
import json; d=json.load(open('gpurun_out/r06t_bench_driver_style.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step','steps','warmup','step_frac_of_bf16_peak')}, r['launch_ms'], r['frac'], r['traffic'], r['traffic_source'], r.get('energy_j'), d['nominal']['ms_per_step'], d['with_exact_shortcuts']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('non-attention ms per step', round(d['ms_per_step'] - 21 * r['launch_ms'], 2))"
    OUT=$PWD/gpurun_out/prof_r06t_nominal; rm -rf $OUT; mkdir -p $OUT
    timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --shape nominal --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
    python tools/summarize_prof.py $OUT gpurun_out/r06t_nominal 2>&1 | tail -3
    head -16 gpurun_out/r06t_nominal_by_launch_shape.csv | cut -c1-200
    ;;
  *) echo "unknown entry $NAME"; exit 2 ;;
esac
