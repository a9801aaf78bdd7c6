"""A/B of the persistent ping-pong GEMM (csrc/gemm8.hip) against the tile kernels of gemm_dma.hip and an fp32 reference.

    python tools/gemm8_check.py [quick|full] [--time]

Per case: the result with dle_gemm8_mode(1) vs dle_gemm8_mode(0) (bit-identical fraction, max difference), both vs
torch.matmul in fp32, a race screen (the new kernel run several times must reproduce itself bit for bit) and HIP-event times.
JSON lines on stdout."""
import json
import os
import sys

os.environ.setdefault("DLE_GEMM_8PH_MIN_ITEMS", "1")          # let the small / ragged cases through the new kernel (read at first use)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                   # noqa: E402
from deeplearningexamples_amd import functional as F, _cabi as C   # noqa: E402

dev = torch.device("cuda", 0)
lib = C.lib()


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def case(name, m, n, k, layout, dtype=torch.bfloat16, bias=False, act=C.ACT_NONE, aux=False, src=False, out_f32=False,
         splitk=1, accumulate=False, colsum=False, do_time=True, scale=1.0):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + m + 7 * n + 13 * k)
    a_kc, b_kc = layout in ("nt",), layout in ("nt",)
    if layout == "nt":                     # forward: A [m, k], B [n, k]
        a = (torch.randn(m, k, device=dev, generator=g) * scale).to(dtype)
        b = (torch.randn(n, k, device=dev, generator=g) * scale).to(dtype)
        a_kc, b_kc = True, True
        ref_a, ref_b = a.float(), b.float().t()
    elif layout == "nn":                   # data gradient: A [m, k], B [k, n]
        a = (torch.randn(m, k, device=dev, generator=g) * scale).to(dtype)
        b = (torch.randn(k, n, device=dev, generator=g) * scale).to(dtype)
        a_kc, b_kc = True, False
        ref_a, ref_b = a.float(), b.float()
    else:                                  # "tn" weight gradient: A [k, m], B [k, n]
        a = (torch.randn(k, m, device=dev, generator=g) * scale).to(dtype)
        b = (torch.randn(k, n, device=dev, generator=g) * scale).to(dtype)
        a_kc, b_kc = False, False
        ref_a, ref_b = a.float().t(), b.float()
    bias_t = torch.randn(n, device=dev, generator=g) if bias else None
    src_t = torch.randn(m, n, device=dev, generator=g).to(dtype) if src else None
    odt = torch.float32 if (out_f32 or splitk > 1) else dtype
    init = torch.randn(m, n, device=dev, generator=g) if accumulate else None

    def run(mode):
        lib.dle_gemm8_mode(mode)
        out = init.clone() if accumulate else torch.empty(m, n, dtype=odt, device=dev)
        aux_t = torch.empty(m, n, dtype=dtype, device=dev) if aux else None
        cs = torch.zeros(n, dtype=torch.float32, device=dev) if colsum else None
        if colsum:
            o = F.gemm_colsum(a, b, m, n, k, src_t, cs, act=act)
            assert o is not None
            out = o
        else:
            F.gemm(a, b, m, n, k, a_kc, b_kc, out=out, bias=bias_t, act=act, aux=aux_t, mask_src=src_t, splitk=splitk,
                   accumulate=accumulate)
        torch.cuda.synchronize()
        return out, aux_t, cs

    o_old, x_old, c_old = run(0)
    o_new, x_new, c_new = run(1)
    # fp32 reference of the product (+ the epilogue)
    ref = ref_a @ ref_b
    if bias:
        ref = ref + bias_t
    pre = ref.clone()
    if act == C.ACT_RELU:
        ref = torch.relu(ref)
    elif act in (C.ACT_GELU, C.ACT_GELU_DAUX):
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    elif act == C.ACT_TANH:
        ref = torch.tanh(ref)
    elif act == C.ACT_RELU_BWD:
        ref = ref * (src_t.float() > 0)
    elif act == C.ACT_ADD:
        ref = ref + src_t.float()
    elif act == C.ACT_MUL:
        ref = ref * src_t.float()
    if accumulate:
        ref = ref + init
    den = ref.abs().max().item() + 1e-30
    rec = {"case": name, "mnk": [m, n, k], "layout": layout, "dtype": str(dtype).split(".")[-1],
           "err_new_vs_fp32": round((o_new.float() - ref).abs().max().item() / den, 6),
           "err_old_vs_fp32": round((o_old.float() - ref).abs().max().item() / den, 6),
           "new_vs_old_maxdiff": (o_new.float() - o_old.float()).abs().max().item(),
           "bit_identical_frac": round((o_new == o_old).float().mean().item(), 6)}
    if aux:
        rec["aux_new_vs_old_maxdiff"] = (x_new.float() - x_old.float()).abs().max().item()
        if act == C.ACT_GELU:
            rec["aux_err_vs_fp32"] = round((x_new.float() - pre).abs().max().item() / (pre.abs().max().item() + 1e-30), 6)
    if colsum:
        cref = o_new.float().sum(0)
        rec["colsum_err_vs_own_output"] = round((c_new - cref).abs().max().item() / (cref.abs().max().item() + 1e-30), 7)
        rec["colsum_new_vs_old"] = round((c_new - c_old).abs().max().item() / (c_old.abs().max().item() + 1e-30), 7)
    # race screen: the new kernel reproduces itself
    same = True
    for _ in range(4):
        o2, x2, c2 = run(1)
        same = same and bool((o2 == o_new).all().item()) and (not aux or bool((x2 == x_new).all().item()))
    rec["self_reproducible"] = same
    if do_time:
        fl = 2.0 * m * n * k
        for mode, key in ((0, "old"), (1, "new")):
            lib.dle_gemm8_mode(mode)
            out = torch.empty(m, n, dtype=odt, device=dev) if not accumulate else init.clone()
            aux_t = torch.empty(m, n, dtype=dtype, device=dev) if aux else None
            cs = torch.zeros(n, dtype=torch.float32, device=dev) if colsum else None
            if colsum:
                fn = lambda: F.gemm_colsum(a, b, m, n, k, src_t, cs, act=act)
            else:
                fn = lambda: F.gemm(a, b, m, n, k, a_kc, b_kc, out=out, bias=bias_t, act=act, aux=aux_t, mask_src=src_t,
                                    splitk=splitk, accumulate=False)
            t = timeit(fn)
            rec["us_" + key] = round(t * 1e6, 1)
            rec["tflops_" + key] = round(fl / t / 1e12, 1)
    lib.dle_gemm8_mode(1)
    print(json.dumps(rec), flush=True)
    return rec


def main():
    full = "full" in sys.argv
    bf, hf = torch.bfloat16, torch.float16
    # ragged / small shapes first (edge tiles, K slices, every epilogue class)
    case("ragged_nt_plain", 1000, 520, 256, "nt", do_time=False)
    case("ragged_nt_bias_gelu_aux", 777, 776, 320, "nt", bias=True, act=C.ACT_GELU, aux=True, do_time=False)
    case("ragged_nt_bias_f32", 515, 1032, 192, "nt", bias=True, out_f32=True, do_time=False)
    case("ragged_nn_add", 900, 520, 384, "nn", act=C.ACT_ADD, src=True, do_time=False)
    case("ragged_nn_relu_bwd_f16", 1031, 264, 128, "nn", dtype=hf, act=C.ACT_RELU_BWD, src=True, do_time=False)
    case("ragged_tn_splitk3", 520, 776, 1536, "tn", splitk=3, do_time=False)
    case("ragged_tn_splitk2_acc", 264, 1000, 2048, "tn", splitk=2, accumulate=True, do_time=False)
    case("tn_f32_nosplit_acc", 512, 768, 640, "tn", out_f32=True, accumulate=True, do_time=False)
    case("nn_mul_colsum", 1024, 512, 256, "nn", act=C.ACT_MUL, src=True, colsum=True, do_time=False)
    case("ktail_nt_bias_relu_f16", 1000, 1024, 480, "nt", dtype=hf, bias=True, act=C.ACT_RELU, do_time=False)
    case("ktail_nn_add", 777, 520, 1000, "nn", act=C.ACT_ADD, src=True, do_time=False)
    case("dlrm_top0_fwd_f16", 65536, 1024, 480, "nt", dtype=hf, bias=True, act=C.ACT_RELU)
    # the layers of the metric workloads
    case("bert_ffn1_fwd", 32768, 4096, 1024, "nt", bias=True, act=C.ACT_GELU_DAUX, aux=True)
    case("bert_plain_fwd", 32768, 4096, 1024, "nt")
    case("bert_ffn2_fwd", 32768, 1024, 4096, "nt", bias=True)
    case("bert_qkv_fwd", 32768, 3072, 1024, "nt", bias=True)
    case("bert_ao_fwd", 32768, 1024, 1024, "nt", bias=True)
    case("bert_ffn2_dgrad_mul_colsum", 32768, 4096, 1024, "nn", act=C.ACT_MUL, src=True, colsum=True)
    case("bert_ffn1_dgrad_add", 32768, 1024, 4096, "nn", act=C.ACT_ADD, src=True)
    case("bert_ao_dgrad", 32768, 1024, 1024, "nn")
    case("bert_qkv_dgrad_add", 32768, 1024, 3072, "nn", act=C.ACT_ADD, src=True)
    case("bert_ffn_wgrad", 4096, 1024, 32768, "tn", splitk=F.pick_splitk(4096, 1024, 32768, 1024))
    case("bert_ffn2_wgrad", 1024, 4096, 32768, "tn", splitk=F.pick_splitk(1024, 4096, 32768, 1024))
    case("bert_qkv_wgrad", 3072, 1024, 32768, "tn", splitk=F.pick_splitk(3072, 1024, 32768, 1024))
    case("bert_ao_wgrad", 1024, 1024, 32768, "tn", splitk=F.pick_splitk(1024, 1024, 32768, 1024))
    case("dlrm_top1_fwd_f16", 65536, 1024, 1024, "nt", dtype=hf, bias=True, act=C.ACT_RELU)
    case("dlrm_top1_dgrad_f16", 65536, 1024, 1024, "nn", dtype=hf, act=C.ACT_RELU_BWD, src=True, colsum=True)
    case("dlrm_top1_wgrad_f16", 1024, 1024, 65536, "tn", dtype=hf, splitk=F.pick_splitk(1024, 1024, 65536, 1024))
    case("sq4096", 4096, 4096, 4096, "nt")
    case("sq8192", 8192, 8192, 8192, "nt")
    if full:
        case("sq8192_nn", 8192, 8192, 8192, "nn")
        case("sq8192_tn", 8192, 8192, 8192, "tn", out_f32=False)
        case("mlm_decoder_f32", 5120, 30528, 1024, "nt", bias=True, out_f32=True)


if __name__ == "__main__":
    main()
