#!/bin/bash
# Matrix-pipe / LDS counters of the round-4 kernels and of the two round-3 kernels the verdict asked for (gemm_expand_kernel,
# gemm_smallm_kernel): rocprofv3 --pmc passes (kernel trace only, own runs) over small probe scripts; per-kernel averages.
# usage (through gpurun): bash tools/pmc_r04_kernels.sh > gpurun_out/r04_pmc_new_kernels.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"
run() {   # <kernel substring> <probe script...>
  sub=$1; shift
  for grp in "$G1" "$G2"; do bash tools/pmc_any.sh "$grp" "$sub" -- python "$@" 2>/dev/null; done
}
echo "== conv3x3_wgrad_kernel (tools/probes/wgrad3x3_bench.py)"; run conv3x3_wgrad_kernel tools/probes/wgrad3x3_bench.py
echo "== wgrad1x1_kernel (tools/probes/wgrad1x1_bench.py)"; run wgrad1x1_kernel tools/probes/wgrad1x1_bench.py
echo "== stem7 kernels (tools/probes/stem_bench.py)"; run stem7_ tools/probes/stem_bench.py
echo "== gemm_expand_kernel / conv_bnload_kernel (ResNet-50 step, tools/replay_step.py rn50)"; run gemm_expand_kernel tools/replay_step.py rn50
run conv_bnload_kernel tools/replay_step.py rn50
echo "== gemm_smallm_kernel (Tacotron2 probe, tools/probes/smallm_policy.py)"; run gemm_smallm_kernel tools/probes/smallm_policy.py
echo "== head_bce_kernel / emb_onehot_kernel / emb_sgd_lists (DLRM step, tools/replay_step.py dlrm)"; run head_bce_kernel tools/replay_step.py dlrm
run emb_onehot_kernel tools/replay_step.py dlrm
run emb_sgd_lists tools/replay_step.py dlrm
echo "== conv_bnbwd_kernel (ResNet-50 step)"; run conv_bnbwd_kernel tools/replay_step.py rn50
