"""GPU parity, round 2: grouped (per-expert) launches and 128-element scales (BASELINE config 5), the module-level
contracts of SURVEY 8c (FlatQuantizedLinear._eval_forward, {Inv,SVD}DecomposeTransMatrix.forward incl. inv_t / diag,
SVDSingleTransMatrix.forward on the HIP path), deploy.nn.Quantizer(lac=True), row shards through the kernel."""
import numpy as np
import pytest

from conftest import BOUND37, flip_ok
from tests.conftest import same_bits
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def rel_err(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


# ------------------------------------------------------------------------------------------------- grouped launches
def _groups(rng, rows, n_groups, empties=True):
    cuts = np.sort(rng.integers(0, rows + 1, size=n_groups - 1))
    offs = np.concatenate([[0], cuts, [rows]]).astype(np.int64)
    if empties and n_groups >= 4:
        offs[1] = offs[0]                      # group 0 empty
        offs[-2] = offs[-1]                    # last group empty
    return offs


@pytest.mark.parametrize("shape,rows,n_groups", [((32, 64), 333, 9), ((64, 112), 61, 5), ((64, 64), 200, 7),
                                                  ((56, 64), 40, 4), ((64, 128), 30, 3), ((32, 64), 4100, 256)])
def test_grouped_quant_stage_bit_exact_and_transform_shared(ops, shape, rows, n_groups):
    """Per-group clip pairs: packed and fake-quant outputs of the grouped launch equal the oracle's quantiser applied,
    group by group, to the transform the SAME launch returns; the transform equals the ungrouped launch's bit for bit."""
    M, N = shape
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, M * N)) * (1 + 5 * (rng.random((rows, 1)) < 0.1))).astype(np.float16)
    L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
    R = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
    offs = _groups(rng, rows, n_groups)
    smax = rng.uniform(0.3, 1.0, n_groups).astype(np.float32)
    smin = rng.uniform(0.3, 1.0, n_groups).astype(np.float32)
    smax[n_groups // 2] = 0.05                                           # one group clips hard: the clamp route
    xd, Ld, Rd = dev(x), dev(L), dev(R)
    y_plain = host(ops.kron_quant(xd, Ld, Rd, flags=T).y)
    for fl in (P | T | R16, F | T | R16):
        o = ops.kron_quant_grouped(xd, Ld, Rd, dev(offs), dev(smax), dev(smin), fl)
        y16 = host(o.y)
        assert np.array_equal(y16, y_plain)
        for gi in range(n_groups):
            a, b = int(offs[gi]), int(offs[gi + 1])
            if a == b:
                continue
            ref = O.quant_outputs(y16[a:b].astype(np.float32), float(smax[gi]), float(smin[gi]))
            if fl & P:
                assert np.array_equal(host(o.q[0])[a:b], ref["packed"]), (gi, a, b)
                assert np.array_equal(host(o.scale[0])[a:b], ref["scale16"])
            else:
                assert same_bits(host(o.fq[0])[a:b], ref["fq"]), (gi, a, b)
    # packed-only launch (its own kernels: one wave per token): same bytes as the packed + transform launch
    o1 = ops.kron_quant_grouped(xd, Ld, Rd, dev(offs), dev(smax), dev(smin), P | R16)
    o2 = ops.kron_quant_grouped(xd, Ld, Rd, dev(offs), dev(smax), dev(smin), P | T | R16)
    assert torch.equal(o1.q[0], o2.q[0]) and torch.equal(o1.scale[0], o2.scale[0])


def test_grouped_edge_cases(ops):
    rng = np.random.default_rng(7)
    x = dev(rng.standard_normal((6, 2048)).astype(np.float16))
    L, R = dev((rng.standard_normal((32, 32)) / 6).astype(np.float16)), dev((rng.standard_normal((64, 64)) / 8).astype(np.float16))
    one = ops.kron_quant(x, L, R, [(0.9, 0.8)], P)
    # empty groups everywhere, one single-row group, all groups with the same pair == the ungrouped launch
    offs = dev(np.array([0, 0, 1, 1, 6, 6, 6], dtype=np.int64))
    sm, sn = dev(np.full(6, 0.9, np.float32)), dev(np.full(6, 0.8, np.float32))
    o = ops.kron_quant_grouped(x, L, R, offs, sm, sn, P)
    assert torch.equal(o.q[0], one.q[0]) and torch.equal(o.scale[0], one.scale[0])
    # no rows at all
    e = ops.kron_quant_grouped(x[:0], L, R, dev(np.array([0, 0], dtype=np.int64)), sm[:1], sn[:1], P)
    assert e.q[0].shape == (0, 1024)
    # a pair only the general MFMA kernel takes (60 x 62) is grouped too; a pair without any kernel is refused
    from flatquant_amd._lib import FqError
    xa = dev(rng.standard_normal((4, 60 * 62)).astype(np.float16))
    La, Ra = dev((rng.standard_normal((60, 60)) / 8).astype(np.float16)), dev((rng.standard_normal((62, 62)) / 8).astype(np.float16))
    ga = ops.kron_quant_grouped(xa, La, Ra, dev(np.array([0, 1, 4], dtype=np.int64)), sm[:2] * 0 + dev(np.array([0.9, 0.6], np.float32)),
                                sn[:2], P)
    for (a, b, s_) in ((0, 1, 0.9), (1, 4, 0.6)):
        ref = ops.kron_quant(xa[a:b].contiguous(), La, Ra, [(s_, 0.8)], P)
        assert torch.equal(ga.q[0][a:b], ref.q[0]) and torch.equal(ga.scale[0][a:b], ref.scale[0])
    with pytest.raises(FqError):
        ops.kron_quant_grouped(dev(np.zeros((4, 200 * 200), np.float16)), dev(np.eye(200, dtype=np.float16)),
                               dev(np.eye(200, dtype=np.float16)), dev(np.array([0, 4], dtype=np.int64)), sm[:1], sn[:1], P)


def test_moe_routed_experts_match_reference_flow(ops, golden):
    """flatquant/model_tools/deepseekv3_utils.py:427-452 with the reference's own primitives (tools/gen_golden.py):
    w1_trans once + shared routed quantiser (64 x 112), then the experts' hidden rows through routed_w2_trans (32 x 64,
    shared and per expert) and the w2 quantiser."""
    g = golden("moe_grouped")
    offs = g["offsets"]
    E = len(offs) - 1
    # routing plumbing: same row order as the reference's torch.where loop
    tok_idx, offsets = ops.moe_group_rows(dev(g["indices"]), E)
    assert np.array_equal(host(offsets), offs) and np.array_equal(host(tok_idx), g["rows_tok"])
    # stage 1: transform + fake-quant of all tokens, gathered per expert
    s1 = (float(g["sig1"][0]), float(g["sig1"][1]))
    o = ops.kron_quant(dev(g["x"]), dev(g["L1"]), dev(g["R1"]), [s1], F | T | R16)
    assert rel_err(host(o.y), g["xt"]) <= 1e-3
    fq1 = host(o.fq[0][tok_idx])
    assert flip_ok(fq1, g["fq1"], "moe flow fq1 vs reference golden", BOUND37) and rel_err(fq1, g["fq1"]) <= 0.08      # (one INT4 step of a 4-bit grid)
    assert same_bits(host(o.fq[0]), O.quant_outputs(host(o.y).astype(np.float32), *s1)["fq"])
    # stage 2: grouped launch, shared transform + the shared quantiser expanded per expert
    s2 = np.tile(g["sig2"][None, :], (E, 1)).astype(np.float32)
    h = dev(g["h"])
    o2 = ops.kron_quant_grouped(h, dev(g["L2"]), dev(g["R2"]), offsets, dev(s2[:, 0]), dev(s2[:, 1]), F | T | R16)
    assert rel_err(host(o2.y), g["y2_shared"]) <= 1e-3
    assert flip_ok(host(o2.fq[0]), g["fq2_shared"], "moe flow fq2 shared vs reference golden", BOUND37)
    # per-expert transforms and quantisers (the routed_w2_trans[i] branch)
    o3 = ops.kron_quant_grouped(h, dev(g["L2e"]), dev(g["R2e"]), offsets, dev(g["sig2e"][:, 0].copy()),
                                dev(g["sig2e"][:, 1].copy()), F | R16)
    assert flip_ok(host(o3.fq[0]), g["fq2_indep"], "moe flow fq2 per-expert vs reference golden", BOUND37)
    ref = O.kron_quant_grouped(g["h"], g["L2e"], g["R2e"], offs, g["sig2e"][:, 0], g["sig2e"][:, 1], round_y_f16=True)
    assert flip_ok(host(o3.fq[0]), ref["fq"], "moe flow fq2 per-expert vs oracle", BOUND37)


# ------------------------------------------------------------------------------------------- 128-element scales
@pytest.mark.parametrize("tag", ["32x64", "64x64", "56x64", "64x112"])
def test_group128_scales(ops, golden, tag):
    g = golden("group128")
    x, L, R = dev(g[f"{tag}_x"]), dev(g[f"{tag}_L"]), dev(g[f"{tag}_R"])
    rows, d = g[f"{tag}_x"].shape
    y = ops.kron_quant(x, L, R, flags=T).y
    assert rel_err(host(y), g[f"{tag}_y"]) <= 1e-3
    for ci in range(2):
        sig = (float(g[f"{tag}_sig{ci}"][0]), float(g[f"{tag}_sig{ci}"][1]))
        ref = O.quant_outputs(host(y).astype(np.float32), *sig, groupsize=128)
        o = ops.kron_quant(x, L, R, [sig], P | R16, groupsize=128)        # fused where N = 64, two launches otherwise
        assert o.scale[0].shape == (rows, d // 128)
        assert np.array_equal(host(o.q[0]), ref["packed"])
        assert np.array_equal(host(o.scale[0]), ref["scale16"])
        o = ops.kron_quant(x, L, R, [sig], F | R16, groupsize=128)
        assert same_bits(host(o.fq[0]), ref["fq"])
        assert flip_ok(host(o.fq[0]), g[f"{tag}_fq{ci}"], f"group128 {tag} clip {ci} vs reference golden", BOUND37)       # vs the reference's vLLM ActivationQuantizer
    if tag in ("32x64", "64x64"):                                         # grouped + 128-element scales in one launch
        offs = dev(np.array([0, 1, 1, rows], dtype=np.int64))
        sm, sn = dev(np.array([0.9, 0.5, 0.7], np.float32)), dev(np.array([0.8, 0.5, 0.6], np.float32))
        o = ops.kron_quant_grouped(x, L, R, offs, sm, sn, P | R16, groupsize=128)
        r0 = O.quant_outputs(host(y)[:1].astype(np.float32), 0.9, 0.8, groupsize=128)
        r2 = O.quant_outputs(host(y)[1:].astype(np.float32), float(np.float32(0.7)), float(np.float32(0.6)), groupsize=128)
        assert np.array_equal(host(o.q[0]), np.concatenate([r0["packed"], r2["packed"]]))
        assert np.array_equal(host(o.scale[0]), np.concatenate([r0["scale16"], r2["scale16"]]))


def test_activation_quantizer_groupsize_module(golden):
    from flatquant_amd.flatquant import ActivationQuantizer
    g = golden("group128")
    q = ActivationQuantizer(bits=4, sym=True, lac=True, groupsize=128).cuda()
    q.clip_factor_a_max.data.fill_(1.7), q.clip_factor_a_min.data.fill_(-0.4)
    fq = q(dev(g["64x112_y"]))
    assert np.array_equal(host(fq), g["64x112_fq1"])                      # same fp16 input -> bit-exact vs the reference


# ------------------------------------------------------------------------------------ module-level contracts (8c)
def test_decompose_trans_matrix_forward_diag_and_inv_t(golden):
    from flatquant_amd.flatquant import InvDecomposeTransMatrix, SVDDecomposeTransMatrix
    g = golden("modules")
    x = dev(g["dec_x"])
    for cls in (InvDecomposeTransMatrix, SVDDecomposeTransMatrix):
        tr = cls(64, 64, add_diag=True)
        sd = {k: torch.from_numpy(g["dec_" + k]) for k in ("matrix_left", "matrix_right", "matrix_left_inv",
                                                          "matrix_right_inv", "diag_scale")}
        tr.load_state_dict(sd)
        tr = tr.cuda()
        y = tr(x)
        assert y.dtype == torch.float16 and rel_err(host(y), g["dec_y"]) <= 1e-3
        yi = tr(x, inv_t=True)                                            # x / diag, inverse-transposed factors
        assert rel_err(host(yi), g["dec_y_inv_t"]) <= 1e-3
        tr.use_diag = False
        assert rel_err(host(tr(x)), g["dec_y_nodiag"]) <= 1e-3
        # against the oracle with the exact op order (diag multiply rounded to fp16 first)
        ref = O.kron_transform(g["dec_x"], g["dec_matrix_left"].astype(np.float16), g["dec_matrix_right"].astype(np.float16))
        assert np.mean(host(tr(x)) != ref.reshape(6, -1).astype(np.float16)) <= 5e-3
        # the fp16 copies of the matrices are made once (stable workspace keys), not per call
        assert len(tr._f16._c) <= 4


def test_single_trans_matrix_forward_on_the_hip_path(golden):
    from flatquant_amd.flatquant import InvSingleTransMatrix, SVDSingleTransMatrix
    g = golden("modules")
    x = dev(g["single_x"])                                                # [T, head_dim, H]: heads last (llama_utils.py:276)
    for cls in (SVDSingleTransMatrix, InvSingleTransMatrix):
        st = cls(32)
        st.load_state_dict({"matrix": torch.from_numpy(g["single_matrix"]), "matrix_inv_t": torch.from_numpy(g["single_matrix_inv_t"])})
        st = st.cuda()
        y = st(x)
        assert y.shape == x.shape and y.dtype == torch.float16
        assert rel_err(host(y), g["single_y"]) <= 1e-3
        assert rel_err(host(st(x, inv_t=True)), g["single_y_inv_t"]) <= 1e-3
        ref = O.single_transform(g["single_x"], g["single_matrix"].astype(np.float16)).astype(np.float16)
        assert np.mean(host(y) != ref) <= 5e-3                            # same fp16 operands, fp32 accumulation
    # 64 heads (Llama-2-70B), row counts of 128 / 96 / 64 / 32 per launch unit, and the torch route for anything else
    rng = np.random.default_rng(3)
    st = SVDSingleTransMatrix(64).cuda()
    for rows in (128 * 3, 96, 64 * 5, 32):
        a = rng.standard_normal((rows, 64)).astype(np.float16)
        ref = O.single_transform(a[None], host(st.matrix).astype(np.float16))[0]
        assert rel_err(host(st(dev(a))), ref) <= 1e-3
    a = rng.standard_normal((40, 64)).astype(np.float16)                  # 40 rows: no kernel geometry -> reference op
    assert rel_err(host(st(dev(a))), O.single_transform(a[None], host(st.matrix).astype(np.float16))[0]) <= 2e-3


def test_flat_quantized_linear_eval_forward(golden):
    """flat_linear.py:75-80: fake-quant of the input (HIP), then the wrapped linear."""
    from types import SimpleNamespace
    from flatquant_amd.flatquant import FlatQuantizedLinear
    g = golden("modules")
    args = SimpleNamespace(w_bits=4, w_asym=False, a_bits=4, a_asym=False, lac=True, a_groupsize=-1, lwc=False)
    lin = torch.nn.Linear(4096, 96, bias=True)
    lin.weight.data, lin.bias.data = torch.from_numpy(g["fql_w"]).float(), torch.from_numpy(g["fql_b"]).float()
    m = FlatQuantizedLinear(args, lin)
    m.act_quantizer.clip_factor_a_max.data.fill_(3.3), m.act_quantizer.clip_factor_a_min.data.fill_(2.1)
    m.reparameterize()
    m = m.half().cuda()
    x = dev(g["fql_x"])
    fq = m.act_quantizer(x)
    assert np.array_equal(host(fq), g["fql_fq"])                          # the quantiser: bit-exact vs the reference module
    out = m(x)
    ref = torch.nn.functional.linear(torch.from_numpy(g["fql_fq"]).float(), torch.from_numpy(g["fql_w"]).float(),
                                     torch.from_numpy(g["fql_b"]).float()).numpy()
    assert out.dtype == torch.float16 and rel_err(host(out), ref) <= 2e-3
    assert rel_err(host(out), g["fql_out"]) <= 4e-3                       # the reference's own fp16 CPU GEMM


def test_deploy_quantizer_lac_bit_exact_vs_reference_module(ops, golden):
    from flatquant_amd import deploy
    g = golden("quantizer_lac")
    for ci in range(3):
        qz = deploy.nn.Quantizer(lac=True).cuda()
        qz.clip_factor_a_max.fill_(float(g[f"clip{ci}"][0])), qz.clip_factor_a_min.fill_(float(g[f"clip{ci}"][1]))
        p = qz(dev(g[f"x{ci}"]))
        assert np.array_equal(host(p.scales_x).reshape(-1), g[f"scales{ci}"])
        assert np.array_equal(host(p.quantized_x), g[f"packed{ci}"])


# -------------------------------------------------------------------------------------------- row shards (8e)
def test_row_shards_through_the_kernel_equal_the_unsharded_launch(ops):
    """sharding.shard_rows: every rank's shard through the real kernel, concatenated, equals one launch over all rows
    bit for bit (rows are independent: no data-path collective is needed or used)."""
    from flatquant_amd import sharding
    rng = np.random.default_rng(11)
    rows = 1000
    x = dev(rng.standard_normal((rows, 4096)).astype(np.float16))
    L, R = dev((rng.standard_normal((64, 64)) / 8).astype(np.float16)), dev((rng.standard_normal((64, 64)) / 8).astype(np.float16))
    full = ops.kron_quant(x, L, R, [(0.98, 0.97)], P | NC0)
    for world in (2, 3, 8):
        qs, ss = [], []
        for rank in range(world):
            a, b = sharding.shard_rows(rows, world, rank)
            o = ops.kron_quant(x[a:b].contiguous(), L, R, [(0.98, 0.97)], P | NC0)
            qs.append(o.q[0]), ss.append(o.scale[0])
        assert torch.equal(torch.cat(qs), full.q[0]) and torch.equal(torch.cat(ss), full.scale[0])


# ------------------------------------------------------------- per-expert matrices in ONE launch (round 3, config 5)
@pytest.mark.parametrize("shape,dtype", [((32, 64), "f16"), ((64, 112), "f16"), ((32, 64), "bf16"), ((64, 64), "f16")])
def test_grouped_launch_with_one_factor_pair_per_group(ops, shape, dtype):
    """fq_kron_quant_grouped_mats_*: routed_w2_trans[i] (deepseekv3_utils.py:443-446) without the host loop. Every group's rows
    equal what the plain launch gives with THAT group's matrices and clip pair, bit for bit (same kernel family, transform and
    fake-quant outputs); empty groups, a 1-row group, 40 groups; the workspace is prepared once and reused."""
    M, N = shape
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rng = np.random.default_rng(M * N)
    rows, G = 700, 40
    x = torch.from_numpy((rng.standard_normal((rows, M * N)) * (1 + 4 * (rng.random((rows, 1)) < 0.1))).astype(np.float32)).to(td).cuda()
    Lg = torch.from_numpy((rng.standard_normal((G, M, M)) / np.sqrt(M)).astype(np.float32)).to(td).cuda()
    Rg = torch.from_numpy((rng.standard_normal((G, N, N)) / np.sqrt(N)).astype(np.float32)).to(td).cuda()
    offs = _groups(rng, rows, G)
    offs[5] = offs[4] + 1 if offs[4] + 1 <= offs[6] else offs[5]          # a 1-row group where the cuts allow it
    smax = rng.uniform(0.3, 1.0, G).astype(np.float32)
    smin = rng.uniform(0.3, 1.0, G).astype(np.float32)
    for fl in (F | T | R16, P | T | R16):
        o = ops.kron_quant_grouped(x, Lg, Rg, dev(offs), dev(smax), dev(smin), fl)
        o_again = ops.kron_quant_grouped(x, Lg, Rg, dev(offs), dev(smax), dev(smin), fl)     # cached images (FQ_WS_PREPARED)
        for g in range(G):
            a, b = int(offs[g]), int(offs[g + 1])
            if b == a:
                continue
            one = ops.kron_quant(x[a:b], Lg[g].contiguous(), Rg[g].contiguous(), [(float(smax[g]), float(smin[g]))], F | T | R16 if fl & F else T)
            if shape != (64, 64):   # (64 x 64: the plain launch runs fq_kron64, whose K-steps take the contraction index in
                                    #  another order — equal up to the last bits of the accumulators, compared below)
                assert torch.equal(o.y[a:b].view(torch.int16), one.y.view(torch.int16)), (g, "transform")
            else:
                assert float((o.y[a:b] != one.y).float().mean()) <= 5e-3
            lowp = "bf16" if dtype == "bf16" else "f16"
            yv = O.bf16_from_bits(o.y[a:b].cpu().view(torch.int16).numpy().view(np.uint16)) if dtype == "bf16" else host(o.y[a:b]).astype(np.float32)
            ref = O.quant_outputs(yv, float(smax[g]), float(smin[g]), round_y_f16=True, lowp=lowp)   # the group's OWN clip pair
            if fl & F:
                if shape != (64, 64):
                    assert torch.equal(o.fq[0][a:b].view(torch.int16), one.fq[0].view(torch.int16)), (g, "fake-quant")
                got = o.fq[0][a:b].cpu().view(torch.int16).numpy().view(np.uint16)
                want = O.bf16_bits(ref["fq"]) if dtype == "bf16" else ref["fq"].view(np.uint16)
                assert np.array_equal(got, want), (g, "fake-quant vs oracle")
            else:   # packed: the quantiser stage on the launch's own (rounded) transform, bit for bit
                assert np.array_equal(host(o.q[0][a:b]), ref["packed"]), (g, "packed")
        assert torch.equal(o.y.view(torch.int16), o_again.y.view(torch.int16))
    # transform only: no clip pairs needed at the C ABI (ops passes them anyway); and the reference fixture's per-expert branch
    if shape == (32, 64) and dtype == "f16":
        import os
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "moe_grouped.npz"))
        o3 = ops.kron_quant_grouped(dev(g["h"]), dev(g["L2e"]), dev(g["R2e"]), dev(g["offsets"]), dev(g["sig2e"][:, 0].copy()),
                                    dev(g["sig2e"][:, 1].copy()), F | R16)
        assert flip_ok(host(o3.fq[0]), g["fq2_indep"], "moe bf16/again fq2 per-expert vs golden", BOUND37)


def test_launch_plans_and_static_output_modules():
    """ops.kron_plan / ops.rowquant_plan (round 4): a prepared launch with static outputs gives the bits of the general entry point,
    call after call, on changing inputs; a wrong shape / dtype / device is refused; deploy.nn.OnlineTrans / Quantizer with
    static_outputs = True return the same values as without (their PackedQuantizedTensor is rewritten by the next call), re-plan
    when a matrix or a clip factor changes in place, and after ops.invalidate_caches()."""
    import torch
    from flatquant_amd import deploy, ops
    from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, FQ_QUANT_F16, FQ_SIG_F16
    gen = torch.Generator().manual_seed(77)
    for (M, N, rows) in ((64, 64, 6), (112, 128, 40), (64, 128, 9)):
        L = (torch.randn(M, M, generator=gen) / M ** 0.5).half().cuda()
        R = (torch.randn(N, N, generator=gen) / N ** 0.5).half().cuda()
        xs = [torch.randn(rows, M * N, generator=gen).half().cuda() for _ in range(3)]
        sigs = [(0.98, 0.97), (0.8, 0.6)]
        plan = ops.kron_plan(xs[0], L, R, sigs, FQ_OUT_PACKED | FQ_NO_CLAMP0)
        for x in xs:
            o = plan.run(x)
            ref = ops.kron_quant(x, L, R, sigs, FQ_OUT_PACKED | FQ_NO_CLAMP0)
            for ci in range(2):
                assert torch.equal(o.q[ci], ref.q[ci]) and torch.equal(o.scale[ci], ref.scale[ci]), (M, N, ci)
        import pytest
        with pytest.raises(ValueError):
            plan.run(xs[0][:-1])
        with pytest.raises(ValueError):
            plan.run(xs[0].to(torch.bfloat16))
    x = torch.randn(5, 4096, generator=gen).half().cuda()
    rp = ops.rowquant_plan(x, [(0.9, 0.9)], FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_SIG_F16)
    for _ in range(2):
        x = torch.randn(5, 4096, generator=gen).half().cuda()
        o, ref = rp.run(x), ops.rowquant(x, [(0.9, 0.9)], FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_SIG_F16)
        assert torch.equal(o.q[0], ref.q[0]) and torch.equal(o.scale[0], ref.scale[0])
    # modules
    ot = deploy.nn.OnlineTrans(4096, trans="matmul", decompose=True, lac=True).cuda()
    ot.left_matrix.copy_((torch.randn(64, 64, generator=gen) / 8).half())
    ot.right_matrix.copy_((torch.randn(64, 64, generator=gen) / 8).half())
    ot.clip_factor_a_max.fill_(3.0), ot.clip_factor_a_min.fill_(2.0)
    qz = deploy.nn.Quantizer(lac=True).cuda()
    qz2 = deploy.nn.Quantizer(input_clip_ratio=0.9).cuda()

    def both(mod, x):
        mod.static_outputs = False
        a = mod(x)
        mod.static_outputs = True
        b = mod(x)
        assert torch.equal(a.quantized_x, b.quantized_x) and torch.equal(a.scales_x, b.scales_x)
        assert a.quantized_x.shape == b.quantized_x.shape and a.scales_x.shape == b.scales_x.shape
        return b

    for _ in range(3):
        x3 = torch.randn(2, 3, 4096, generator=gen).half().cuda()
        both(ot, x3), both(qz, x3.reshape(6, 4096)), both(qz2, x3.reshape(6, 4096))
    first = both(ot, x3)
    ptr = first.quantized_x.data_ptr()
    assert both(ot, x3).quantized_x.data_ptr() == ptr                  # static buffers
    ot.left_matrix.copy_((torch.randn(64, 64, generator=gen) / 8).half())   # in-place update: a new plan, right values
    both(ot, x3)
    ot.clip_factor_a_max.fill_(1.0)
    both(ot, x3)
    ot.right_matrix.data.copy_((torch.randn(64, 64, generator=gen) / 8).half())   # through .data: invisible until invalidate_caches()
    ops.invalidate_caches()
    both(ot, x3)
    both(ot, torch.randn(1, 7, 4096, generator=gen).half().cuda())     # another shape: another plan
    # Linear4bit at decode sizes
    from oracle import fq_oracle as O
    import numpy as np
    lin = deploy.nn.Linear4bit(512, 1024, bias=True).cuda()
    lin.weight.copy_(torch.from_numpy(O.pack_i4(torch.randint(-8, 8, (1024, 512), generator=gen, dtype=torch.int32).numpy())))
    lin.weight_scales.copy_(torch.rand(1024, 1, generator=gen) * 0.02 + 0.001)
    lin.bias.copy_(torch.randn(1024, generator=gen).half())
    for rows in (4, 4, 16):
        pk = deploy.PackedQuantizedTensor(torch.from_numpy(O.pack_i4(torch.randint(-8, 8, (rows, 512), generator=gen, dtype=torch.int32).numpy())).cuda()
                                          .reshape(1, rows, 256), (torch.rand(1, rows, 1, generator=gen) * 0.05 + 0.001).half().cuda())
        lin.static_outputs = False
        a = lin(pk)
        lin.static_outputs = True
        b = lin(pk)
        assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16)), rows
    lin.weight_scales.mul_(2.0)            # in-place update: a new plan
    lin.static_outputs = False
    a = lin(pk)
    lin.static_outputs = True
    assert torch.equal(a.view(torch.int16), lin(pk).view(torch.int16))


def test_prepared_fragment_images_are_shared_across_streams_and_graph_capture():
    """The fragment image of a factor pair prepared on one stream serves every other stream once its preparing launch has completed
    (round 4): a launch captured into a HIP graph after a warm-up on the default stream must not carry its own prepare kernel. Checked
    through the cache itself (same image tensor, `prepared`) and through the results (another stream, a captured graph)."""
    import torch
    from flatquant_amd import ops
    from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
    gen = torch.Generator().manual_seed(123)
    M, N = 64, 112
    L = (torch.randn(M, M, generator=gen) / 8).half().cuda()
    R = (torch.randn(N, N, generator=gen) / 10).half().cuda()
    x = torch.randn(33, M * N, generator=gen).half().cuda()
    ops.invalidate_caches()
    ref = ops.kron_quant(x, L, R, [(0.98, 0.97)], FQ_OUT_PACKED | FQ_NO_CLAMP0)
    torch.cuda.synchronize()
    ws0 = ops._kron_workspace(x.device, M, N, L, R)
    assert ws0[2] is True
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ws1 = ops._kron_workspace(x.device, M, N, L, R)
        assert ws1[2] is True and ws1[0].data_ptr() == ws0[0].data_ptr()
        o = ops.kron_quant(x, L, R, [(0.98, 0.97)], FQ_OUT_PACKED | FQ_NO_CLAMP0)
    side.synchronize()
    assert torch.equal(o.q[0], ref.q[0]) and torch.equal(o.scale[0], ref.scale[0])
    # a pair first seen on the default stream, then captured
    L2 = (torch.randn(M, M, generator=gen) / 8).half().cuda()
    out = ops.kron_quant(x, L2, R, [(0.98, 0.97)], FQ_OUT_PACKED | FQ_NO_CLAMP0)
    want_q, want_s = out.q[0].clone(), out.scale[0].clone()
    torch.cuda.synchronize()
    assert ops.images_ready() == 0     # (round 5: inside a capture no event may be asked; completion is established here)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        assert ops._kron_workspace(x.device, M, N, L2, R)[2] is True      # no prepare kernel inside the capture
        cap = ops.kron_quant(x, L2, R, [(0.98, 0.97)], FQ_OUT_PACKED | FQ_NO_CLAMP0)
    cap.q[0].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap.q[0], want_q) and torch.equal(cap.scale[0], want_s)


def test_captured_launch_keeps_its_fragment_image_alive_through_invalidate_and_eviction():
    """ A launch issued under stream capture is handed a fragment image the cache owns; the graph replays that raw pointer.
    invalidate_caches() — also fired by the load_state_dict hook of ANY module — and LRU eviction must not free it: the image is
    pinned (ops._WS_PINNED). Here: warm-up, images_ready(), capture (the captured launch shares the warm-up's image: no prepare
    kernel in the graph), then the caches are dropped, the allocator is flooded with same-sized blocks, and the replay must still
    reproduce the eager result bit for bit. Also: under capture an image whose completion is unknown is NOT shared."""
    from flatquant_amd import ops
    rng = np.random.RandomState(81)
    M, N = 64, 128
    left = torch.from_numpy((rng.randn(M, M) / 8).astype(np.float16)).cuda()
    right = torch.from_numpy((rng.randn(N, N) / 11).astype(np.float16)).cuda()
    x = torch.from_numpy(rng.randn(256, M * N).astype(np.float16)).cuda()
    ops.invalidate_caches()
    want = ops.kron_quant(x, left, right)
    torch.cuda.synchronize()
    assert ops.images_ready() == 0
    n_lru = len(ops._WS_LRU)
    q = torch.empty_like(want.q[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = ops.kron_quant(x, left, right)
        q.copy_(o.q[0])
    assert len(ops._WS_PINNED) >= 1                       # the image the graph was handed
    assert len(ops._WS_LRU) == n_lru + 1                  # shared under the capture stream's key, not prepared again
    ops.invalidate_caches()
    junk = [torch.full_like(w[0], 0xFF) for w in ops._WS_PINNED.values() for _ in range(8)]   # would recycle a freed image's block
    for _ in range(3):
        q.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(q, want.q[0])
    del junk
    # completion unknown -> no sharing inside a capture (the launch prepares its own image in the graph)
    ops.invalidate_caches()
    left2 = left.clone()
    ops.kron_quant(x, left2, right)                       # eager: registers an image whose event has not been asked yet
    ent = next(iter(ops._WS_ANY.values()))
    ent[3][0] = False
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        assert ops._ws_shared((x.device.index, 12345, M, N, left2.data_ptr(), left2._version, right.data_ptr(), right._version)) is None
