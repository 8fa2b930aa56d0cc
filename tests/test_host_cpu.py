"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the
header declares, argument validation works without a GPU, and the host compiler
(HRNetProgram) reproduces the reference graph's census."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from udp_pose_amd import _lib, hrnet_plan, synth
from udp_pose_amd.config import load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "udp_pose_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(udp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libudp_pose_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS)
    assert lib.udp_abi_version() == _lib.ABI_VERSION == 19


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    assert lib.udp_decode_gaussian(None, 1, 17, 64, 48, None, None, 1, 1, None, None, None, None, None) == -1
    assert b"null pointer" in lib.udp_last_error()
    assert lib.udp_flip_fuse(None, None, None, None, 1, 1, 1, 1, None, None) == -1
    assert lib.udp_mse_loss(None, None, None, 1, 1, 1, 0, None, None, None) == -1
    buf = (C.c_float * 4)()
    assert lib.udp_gaussian_taps_host(4, buf) == -1          # even kernel size
    assert lib.udp_gaussian_taps_host(17, buf) == -1
    h = C.c_void_p()
    assert lib.udp_hrnet_create(None, 0, None, 0, None, 0, 0, 256, 192, 17, C.byref(h)) == -1
    assert lib.udp_hrnet_workspace_bytes(None, 4, 0) == 0
    assert lib.udp_hrnet_destroy(None) == 0


def test_taps_match_documented_opencv_rule():
    buf = (C.c_float * 7)()
    assert _lib.lib().udp_gaussian_taps_host(7, buf) == 0
    np.testing.assert_array_equal(np.frombuffer(buf, np.float32),
                                  np.array([0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125], np.float32))
    buf = (C.c_float * 15)()
    assert _lib.lib().udp_gaussian_taps_host(15, buf) == 0
    t = np.frombuffer(buf, np.float32)
    x = np.arange(15) - 7.0
    ref = np.exp(-x * x / (2 * 2.6 ** 2))
    np.testing.assert_allclose(t, ref / ref.sum(), rtol=1e-6)
    assert abs(float(t.sum()) - 1.0) < 1e-6


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16x2"])
def test_program_census_matches_survey(dtype, monkeypatch):
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0)
    monkeypatch.setenv("UDP_POSE_NO_L1_CONCAT", "1")
    plain = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, dtype)
    monkeypatch.delenv("UDP_POSE_NO_L1_CONCAT")
    prog = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, dtype)
    kinds = [d[1] for d in prog.describe()]
    assert kinds.count(_lib.UDP_OP_STEM) == 1
    n_block = kinds.count(_lib.UDP_OP_BLOCK)       # bf16: the 32 BasicBlocks of the 32-channel branch are one launch each
    assert n_block == (32 if dtype == "bf16" else 0)
    # 294 convs; layer1.0's conv3 and projection shortcut are ONE conv over the concatenated channels
    # split fp16: conv3 of layer1.k and conv1 of layer1.k+1 are one chained launch (udp_conv_op.chain_cout), k = 0..2
    chained = sum(1 for o in prog.ops_array() if o.chain_cout)
    assert chained == (3 if dtype == "f16x2" else 0)
    # merged-launch programs (bf16 / split fp16): the stride-2 convs that end the terms of an exchange-unit output i >= 2 are
    # ONE conv over the concatenated inputs: 4 x (2 -> 1) in stage 3, 2 x ((2 -> 1) + (3 -> 1)) in stage 4
    fused_terms = 0 if dtype == "f32" else 10
    assert kinds.count(_lib.UDP_OP_STEM) + kinds.count(_lib.UDP_OP_CONV) + 2 * n_block + chained == 293 - fused_terms
    monkeypatch.setenv("UDP_POSE_NO_FUSE_CONCAT", "1")
    per_term = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, dtype)
    monkeypatch.delenv("UDP_POSE_NO_FUSE_CONCAT")
    assert [d[1] for d in per_term.describe()].count(_lib.UDP_OP_CONV) == kinds.count(_lib.UDP_OP_CONV) + fused_terms
    assert per_term.macs_per_image() == prog.macs_per_image()
    if fused_terms:
        wide = sorted(d[4:6] for d in prog.describe() if d[3] == 2 and d[4] in (96, 224))
        assert wide == [(96, 128)] * 6 + [(224, 256)] * 2
    assert [d[1] for d in plain.describe()].count(_lib.UDP_OP_CONV) == kinds.count(_lib.UDP_OP_CONV) + 1
    assert plain.macs_per_image() == prog.macs_per_image()
    # the shortcut map is neither written nor read back; a chained conv does not read the 256-channel map again
    assert prog.activation_elems_per_image() == plain.activation_elems_per_image() - 2 * 256 * 64 * 48
    monkeypatch.setenv("UDP_POSE_NO_L1_CHAIN", "1")
    unchained = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, dtype)
    assert not any(o.chain_cout for o in unchained.ops_array())
    assert unchained.macs_per_image() == prog.macs_per_image()
    assert prog.activation_elems_per_image() == unchained.activation_elems_per_image() - chained * 256 * 64 * 48
    cat = [d for d in prog.describe() if d[0] == "layer1.0.conv3"][0]
    assert cat[4:6] == (128, 256)
    assert prog.macs_per_image() == 7670857728                                        # 7.671 GMAC
    ops = prog.ops_array()
    assert ops[len(ops) - 1].out_buf == _lib.UDP_BUF_OUTPUT and ops[len(ops) - 1].cout == 17
    # no op reads and writes the same buffer; every up-sampled addend is lower resolution
    for o in ops:
        ins = [o.in_buf, o.res_buf] + [o.up_buf[u] for u in range(o.n_up)]
        assert o.out_buf not in [b for b in ins if b >= 0]
        assert all(1 <= o.up_shift[u] <= 3 for u in range(o.n_up))
    blob = prog.weight_blob()
    assert blob.nbytes % 256 == 0 and blob.nbytes > (57e6 if dtype == "bf16" else 114e6)


def test_program_rejects_bad_configs():
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0)
    with pytest.raises(ValueError, match="multiple of 32"):
        hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 250, 192, "f32")
    bad = {k: dict(v) if isinstance(v, dict) else v for k, v in synth.W32_EXTRA.items()}
    bad["STAGE3"]["NUM_CHANNELS"] = [32, 64]
    with pytest.raises(ValueError, match="NUM_BRANCHES"):
        synth.hrnet_param_shapes(bad)


def test_yaml_config_defaults(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text("MODEL:\n  TARGET_TYPE: offset\n  IMAGE_SIZE: [192, 256]\n  EXTRA:\n    FINAL_CONV_KERNEL: 1\nTEST:\n  POST_PROCESS: true\n")
    cfg = load_config(str(p))
    assert cfg.MODEL.TARGET_TYPE == "offset" and cfg.LOSS.KPD == 4.0 and cfg.TEST.POST_PROCESS is True
    assert cfg["MODEL"]["EXTRA"]["FINAL_CONV_KERNEL"] == 1 and cfg.MODEL.NUM_JOINTS == 17


def test_lanes_order_every_buffer_hazard():
    """Branch lanes may overlap on the GPU: every RAW / WAR / WAW pair on a physical buffer must be
    ordered by (previous op on the same lane) + (the op's cross-lane wait list), transitively."""
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0)
    ops = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, "bf16").ops_array()
    n = len(ops)
    hb = [0] * n                      # bitset of ops that happen before op i
    last = {}
    for i, o in enumerate(ops):
        m = 0
        preds = [o.wait_op[k] for k in range(o.n_wait)] + ([last[o.lane]] if o.lane in last else [])
        for j in preds:
            assert j < i
            m |= hb[j] | (1 << j)
        hb[i] = m
        last[o.lane] = i
    assert {o.lane for o in ops} == {0, 1, 2, 3} and ops[0].lane == 0 and ops[n - 1].lane == 0
    touched = {}
    for i, o in enumerate(ops):
        reads = [b for b in [o.in_buf, o.res_buf] + [o.up_buf[u] for u in range(o.n_up)] if b >= 0]
        writes = [o.out_buf] if o.out_buf >= 0 else []
        for b in reads:
            for j, kind in touched.get(b, []):
                assert kind == "r" or (hb[i] >> j) & 1, "unordered RAW on buffer %d: op %d -> %d" % (b, j, i)
        for b in writes:
            for j, kind in touched.get(b, []):
                assert (hb[i] >> j) & 1, "unordered WA%s on buffer %d: op %d -> %d" % (kind.upper(), b, j, i)
        for b in reads:
            touched.setdefault(b, []).append((i, "r"))
        for b in writes:
            touched.setdefault(b, []).append((i, "w"))


def test_rsn_program_second_outputs_keep_every_hazard_ordered():
    """RSN-18 in split fp16: the bottleneck's element-wise sums ride in conv epilogues (udp_conv_op.n_out2: the conv also
    writes out + addend slices; out_buf may be UDP_BUF_NONE).  No UDP_OP_FUSE is left inside the bottlenecks, every
    second output / addend has a live buffer, and every RAW / WAR / WAW pair on a physical buffer -- second outputs and
    their addends included -- is ordered by lane order + wait lists."""
    from udp_pose_amd import rsn_plan
    sd = synth.synth_rsn18_state_dict(51, seed=4)
    prog = rsn_plan.RSNProgram(sd, 256, 192, "f16x2")
    ops = prog.ops_array()
    n = len(ops)
    assert sum(o.n_out2 for o in ops) == 48 and sum(1 for o in ops if o.kind == _lib.UDP_OP_FUSE) == 0
    assert sum(1 for o in ops if o.out_buf == _lib.UDP_BUF_NONE) == 24
    hb, last = [0] * n, {}
    for i, o in enumerate(ops):
        m = 0
        for j in [o.wait_op[k] for k in range(o.n_wait)] + ([last[o.lane]] if o.lane in last else []):
            assert j < i
            m |= hb[j] | (1 << j)
        hb[i] = m
        last[o.lane] = i
    touched = {}
    for i, o in enumerate(ops):
        reads = [b for b in [o.in_buf, o.res_buf] + [o.up_buf[u] for u in range(o.n_up)] + [o.add2_buf[k] for k in range(o.n_out2)] if b >= 0]
        writes = ([o.out_buf] if o.out_buf >= 0 else []) + [o.out2_buf[k] for k in range(o.n_out2)]
        assert o.n_out2 == 0 or (o.kind == _lib.UDP_OP_CONV and o.wfmt == 1)
        assert len(set(writes)) == len(writes)
        assert o.ks != 3 or not (set(writes) & set(reads)), "op %d: a 3x3 conv writes a buffer it reads" % i
        for b in reads:
            for j, kind in touched.get(b, []):
                assert kind == "r" or (hb[i] >> j) & 1, "unordered RAW on buffer %d: op %d -> %d" % (b, j, i)
        for b in writes:
            for j, kind in touched.get(b, []):
                assert (hb[i] >> j) & 1, "unordered WA%s on buffer %d: op %d -> %d" % (kind.upper(), b, j, i)
        for b in reads:
            touched.setdefault(b, []).append((i, "r"))
        for b in writes:
            touched.setdefault(b, []).append((i, "w"))


def test_psa_program_emission():
    """pose_hrnet_psa: five ops per BasicBlock (pool, mlp, scale, theta conv, sp) between conv1 and conv2."""
    extra = synth.scaled_extra(32, modules=(1, 2, 1), blocks=2)
    sd = synth.synth_state_dict(extra, 17, "gaussian", seed=6, psa=True)
    prog = hrnet_plan.HRNetProgram(sd, extra, 128, 96, "f32")
    plain = hrnet_plan.HRNetProgram({k: v for k, v in sd.items() if ".deattn." not in k}, extra, 128, 96, "f32")
    n_blocks = sum(1 for k in sd if k.endswith(".deattn.conv_q_right.weight"))
    assert n_blocks == 2 * (2 * 1 + 3 * 2 + 4 * 1)
    assert len(prog.describe()) == len(plain.describe()) + 5 * n_blocks
    kinds = [d[1] for d in prog.describe()]
    for k in (_lib.UDP_OP_PSA_POOL, _lib.UDP_OP_PSA_MLP, _lib.UDP_OP_PSA_SCALE, _lib.UDP_OP_PSA_SP):
        assert kinds.count(k) == n_blocks
    arr = prog.ops_array()
    for i, d in enumerate(prog.describe()):
        if d[1] == _lib.UDP_OP_PSA_SP:
            assert arr[i].n_up == 1 and arr[i].cin * 2 == arr[i].cout
            assert arr[i - 1].kind == _lib.UDP_OP_CONV and arr[i - 1].cout == arr[i].cin      # theta
    with pytest.raises(ValueError):
        sd48 = synth.synth_state_dict(synth.scaled_extra(48, modules=(1, 1, 1), blocks=1), 17, "gaussian", seed=6, psa=True)
        hrnet_plan.HRNetProgram(sd48, synth.scaled_extra(48, modules=(1, 1, 1), blocks=1), 128, 96, "f32")


def test_launch_groups_are_independent():
    """Ops sharing a group id are merged into one launch: no member may read or overwrite what another
    member writes, and a buffer a member reads for the last time is not recycled inside the group."""
    sd = synth.synth_state_dict(synth.W32_EXTRA, 17, "gaussian", seed=0)
    prog = hrnet_plan.HRNetProgram(sd, synth.W32_EXTRA, 256, 192, "bf16")
    ops = prog.ops_array()
    groups = {}
    for i, o in enumerate(ops):
        if o.group:
            groups.setdefault(o.group, []).append(i)
    blocks = {g: i for g, i in groups.items() if ops[i[0]].ks == 3}
    assert len(blocks) == 56 and sorted(set(map(len, blocks.values()))) == [2, 3]       # BasicBlock rows of stage 3 / 4
    assert all(2 <= len(i) <= 4 for i in groups.values())
    for g, idx in groups.items():
        assert idx == list(range(idx[0], idx[0] + len(idx)))                      # consecutive
        outs = [ops[i].out_buf for i in idx]
        assert len(set(outs)) == len(outs)
        for i in idx:
            o = ops[i]
            assert o.kind == _lib.UDP_OP_CONV and o.ks == ops[idx[0]].ks and o.stride == 1 and o.n_up == 0
            reads = {o.in_buf, o.res_buf} - {_lib.UDP_BUF_NONE}
            assert not (reads & (set(outs) - {o.out_buf})), (g, i)


def test_merged_launch_dispatch_order_is_a_permutation():
    """ws_order + the per-8 lookup table (csrc/conv.hip): whatever the member sizes -- the W32 / W48 / RSN groups at
    several batch sizes, counts that are not multiples of 8 (segment-search fallback), one cout block or many --
    every (member, tile, cout block) is computed by exactly one workgroup, the deepest member's workgroups all
    sit in the first chip-wide round, and the launch ends on the shallowest member."""
    lib = _lib.lib()
    rng = np.random.default_rng(7)
    cases = [([64, 256, 512, 1024], [2, 1, 1, 1], [4, 4, 2, 1]), ([256, 512, 1024], [1, 1, 1], [4, 2, 1]),
             ([512, 1024], [1, 1], [2, 1]), ([16, 64, 128, 256], [2, 1, 1, 1], [4, 4, 2, 1]),
             ([3, 10, 19, 38], [2, 1, 1, 1], [4, 4, 2, 1]), ([8, 8], [1, 3], [4, 1]), ([1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1])]
    for _ in range(40):
        n = int(rng.integers(2, 5))
        mult = 8 if rng.random() < 0.6 else 1
        cases.append(([int(rng.integers(1, 200)) * mult for _ in range(n)], [int(rng.integers(1, 5)) for _ in range(n)],
                      [int(rng.choice([1, 2, 4])) for _ in range(n)]))
    seen_table = seen_search = False
    for tiles, ncby, code in cases:
        n = len(tiles)
        total = sum(t * c for t, c in zip(tiles, ncby))
        om, ot, oc = (np.zeros(total, np.uint32) for _ in range(3))
        used = C.c_int(0)
        got = lib.udp_debug_multi_order((C.c_uint * n)(*tiles), (C.c_uint * n)(*ncby), (C.c_int * n)(*code), n, total,
                                        om.ctypes.data, ot.ctypes.data, oc.ctypes.data, C.byref(used))
        assert got == total, (tiles, ncby)
        seen_table |= bool(used.value)
        seen_search |= not used.value
        want = sorted((j, t, c) for j in range(n) for t in range(tiles[j]) for c in range(ncby[j]))
        assert sorted(zip(om.tolist(), ot.tolist(), oc.tolist())) == want, (tiles, ncby, code)
        if total > 512 and tiles[0] * ncby[0] <= 256 and tiles[-1] * ncby[-1] >= total // 3 and used.value:
            assert np.all(np.nonzero(om == 0)[0] < 512 + 64), (tiles, ncby)      # deepest member: first round
            assert om[-1] == n - 1                                             # the launch ends on short workgroups
    assert seen_table and seen_search


def test_bench_gpus_n_launches_its_own_ranks():
    """`bench.py --gpus 2` from a plain shell (no torch.distributed environment) starts two ranks of itself
    before touching the GPU; in this GPU-less container both stop at the explicit "needs a GPU" exit and the
    parent passes the failure on (RSN/exps/RSN18.coco/test.py:158 launches its ranks the same way)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        # both ranks print it unless the launcher, seeing the first rank fail, stops the second before it gets there
        assert 1 <= (r.stdout + r.stderr).count("bench.py needs a GPU") <= 2


def test_f16x2_host_encoding_and_fragment_major_weights():
    """Split-fp16 host helpers (udp-pose_amd/f16x2.py): hi + lo * 2^-11 keeps 22 significant bits over fp16's
    normal range (and degrades gracefully below it); pack_weights_ws puts w[tap][cout][cin] where
    include/udp_pose_hip.h (udp_conv_op.wfmt = 1) says the weight-stationary kernel reads it."""
    import torch
    from udp_pose_amd import f16x2
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-4, 4, 4096)
    x[0], x[1] = 0.0, -0.0
    r = f16x2.decode(f16x2.encode(x[None, :]))[0]
    rel = ((r - x).abs() / x.abs().clamp_min(1e-30))[x.abs() > 2.0 ** -14]
    assert float(rel.max()) <= 2.0 ** -21                      # <= 2^-22 plus the fp32 arithmetic of this check
    assert float((r - x).abs()[x.abs() <= 2.0 ** -14].max()) <= 2.0 ** -34
    # fragment-major layout: 3x3, 40 real couts padded to 64, 48 cin padded to 64
    taps, cout_pad, cin = 9, 64, 48
    w = torch.randn(taps, cout_pad, cin, generator=g) * 0.05
    packed, wexp = f16x2.pack_weights_ws(w)
    packed = packed.view(torch.float16).reshape(taps, 2, 2, 2, 2, 64, 8)   # tap, chunk, pair, nb, plane, lane, j
    # stored scaled by 2^wexp (largest magnitude in [2^13, 2^14)): hi = fp16(w'), lo = fp16(w' - hi), the plain residual
    assert wexp == f16x2.weight_exponent(w) and 2.0 ** 13 <= float(w.abs().max()) * 2.0 ** wexp < 2.0 ** 14
    ws_ = torch.nn.functional.pad(w, (0, 16)) * 2.0 ** wexp
    hi = ws_.to(torch.float16)
    enc = torch.stack([hi, (ws_ - hi.float()).to(torch.float16)], dim=2)     # [tap, cout, plane, k]
    assert float(((enc[:, :, 0].float() + enc[:, :, 1].float()) * 2.0 ** -wexp - torch.nn.functional.pad(w, (0, 16))).abs().max()) <= 2.0 ** -22 * float(w.abs().max())
    assert f16x2.weight_exponent(torch.zeros(3)) == 0 and f16x2.weight_exponent(torch.tensor([1.0])) == 13
    rng = np.random.default_rng(1)
    for _ in range(200):
        tap, c, pair, nb, plane, lane, j = (int(rng.integers(n)) for n in (taps, 2, 2, 2, 2, 64, 8))
        li, kg = lane & 15, lane >> 4
        cout = 32 * pair + 8 * (li >> 2) + 4 * nb + (li & 3)
        assert packed[tap, c, pair, nb, plane, lane, j] == enc[tap, cout, plane, 32 * c + 8 * kg + j]
    assert packed.numel() * 2 == taps * 2 * 2 * 4096
