"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/nerf_fl_amd.h declares, and its host-only entry points (plan builder,
size queries, argument validation) behave.  No kernel is launched."""
import ctypes as C
import os
import re

import pytest

from nerf_fl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "nerf_fl_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nfl_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(L, name)


def test_version_and_errors(L):
    assert L.nfl_abi_version() == _lib.NFL_ABI_VERSION
    assert b"gfx950" in L.nfl_version()
    assert L.nfl_strerror(0) == b"ok"
    assert L.nfl_strerror(-1) != L.nfl_strerror(-4)


@pytest.mark.parametrize("nx,a,t,count", [(10, 0, 0, 595844), (10, 1, 1, 687113)])
def test_param_count_matches_reference(L, nx, a, t, count):
    # 595 844 (coarse / base fine), 687 113 (NeRF-W fine)
    d = _lib.FieldDesc(nx, 4, a, 48, t, 16, 0.1, 0)
    assert L.nfl_param_count(C.byref(d)) == count


def test_plan_structure(L):
    import numpy as np
    d = _lib.FieldDesc(10, 4, 1, 48, 1, 16, 0.1, 0)
    n = L.nfl_plan_bytes(C.byref(d))
    buf = C.create_string_buffer(n)
    assert L.nfl_plan_build(C.byref(d), _lib.NFL_PREC_F16X3, buf, n) == 0
    assert L.nfl_plan_build(C.byref(d), _lib.NFL_PREC_F16X3, buf, 16) == -4          # NFL_ESMALL
    hdr = np.frombuffer(buf.raw[:96], dtype=np.int32)
    magic, prec, nsplit, elem, is_bwd, flags, nx, nkp, has_a, has_t = hdr[:10]
    assert magic == 0x4E464C31 and nsplit == 3 and nx == 10 and nkp == 4 and has_a == 1 and has_t == 1
    assert elem == 0 and is_bwd == 0 and flags == 0
    n_rt, n_rt_sigma, n_rt_static, n_chunks, n_chunks_sigma, n_chunks_static, total_ks, ks_bytes = hdr[12:20]
    assert (n_rt_sigma, n_rt_static, n_rt) == (65, 70, 87)          # no tiles for xyz_encoding_final: folded into dir / transient layer 0
    assert (n_chunks_sigma, n_chunks_static, n_chunks) == (61, 66, 77)
    assert ks_bytes == 2048
    # k-steps: L1 8x4, six 256x256 layers 8x16, skip layer 8x20, sigma 16, dir 4x21, rgb 8,
    # transient 4x17 + 3x(4x8) + 8
    assert total_ks == 32 + 6 * 128 + 160 + 16 + 84 + 8 + 68 + 96 + 8
    assert L.nfl_packed_bytes(C.byref(d), _lib.NFL_PREC_F16X3) == total_ks * 2048 + n_rt * 128
    assert L.nfl_packed_bytes(C.byref(d), _lib.NFL_PREC_F16) == total_ks * 1024 + n_rt * 128


def test_bwd_plan_structure(L):
    import numpy as np
    d = _lib.FieldDesc(10, 4, 1, 48, 1, 16, 0.1, 0)
    n = L.nfl_plan_bytes(C.byref(d))
    buf = C.create_string_buffer(n)
    assert L.nfl_bwd_plan_build(C.byref(d), 0, _lib.NFL_PREC_F16, buf, n) == 0
    hdr = np.frombuffer(buf.raw[:96], dtype=np.int32)
    # dgrad stream of the default backward (prec = NFL_PREC_F16): hi fragments only, no rays-gradient tiles
    assert hdr[1] == 1 and hdr[2] == 1 and hdr[3] == 0 and hdr[4] == 1 and hdr[5] == (4 << 8)      # flags: no rays tiles | n_emb_dir << 8
    n_rt, n_chunks, total_ks = hdr[12], hdr[15], hdr[18]
    # transposed row tiles: transient 4+12+1, rgb^T 4, appearance rows 2, h8 8 (dir' | t0' | sigma), 7 trunk layers x 8; the tiles of
    # the 4- and 8-tile groups travel two per chunk, the latent rows one per chunk
    assert n_rt == 17 + 4 + 2 + 8 + 56
    assert n_chunks == (2 + 6 + 1) + 2 + 2 + 4 + 28
    assert hdr[16] == 9                                     # first chunk of the non-transient part (n_chunks_sigma re-used)
    assert total_ks == 4 * 3 + 12 * 8 + 8 + 4 * 1 + 2 * 8 + 8 * 17 + 56 * 16
    assert L.nfl_bwd_packed_bytes(C.byref(d), 0, _lib.NFL_PREC_F16) == total_ks * 1024 + n_rt * 128
    # with the gradient w.r.t. the rays: + direction rows (1 tile, 8 ks) + encoded-position rows of layers 5 and 1 (2 x 2 tiles, 16 ks)
    assert L.nfl_bwd_plan_build(C.byref(d), 1, _lib.NFL_PREC_F16, buf, n) == 0
    hdr2 = np.frombuffer(buf.raw[:96], dtype=np.int32)
    assert hdr2[5] == (1 | 4 << 8) and hdr2[12] == n_rt + 5 and hdr2[15] == n_chunks + 5 and hdr2[18] == total_ks + 8 + 4 * 16
    # exact-weight chain (F16W) and three-product backward (F16X3): the same tiles as hi + lo fragments (2 KiB per k-step),
    # one tile per chunk; `prec` tells the dgrad kernels apart
    for bp in (_lib.NFL_PREC_F16W, _lib.NFL_PREC_F16X3):
        assert L.nfl_bwd_plan_build(C.byref(d), 0, bp, buf, n) == 0
        hdr3 = np.frombuffer(buf.raw[:96], dtype=np.int32)
        assert hdr3[1] == bp and hdr3[2] == 3 and hdr3[4] == 1 and hdr3[12] == n_rt and hdr3[15] == n_rt and hdr3[18] == total_ks
        assert hdr3[16] == 17                               # transient part: 17 tiles = 17 chunks
        assert L.nfl_bwd_packed_bytes(C.byref(d), 0, bp) == total_ks * 2048 + n_rt * 128
    assert L.nfl_bwd_plan_build(C.byref(d), 0, 7, buf, n) == -1
    # stash sizes: per 32-sample segment 178 / 173 KiB (no slot for xyz_encoding_final's output or its gradient: composed) (+ 4 KiB tail pad), then 84 relu-mask words x 256 B per segment
    assert L.nfl_act_stash_bytes(C.byref(d), 8, 128, _lib.NFL_PREC_F16) == 8 * 4 * 178 * 1024 + 4096 + 8 * 4 * 84 * 256
    assert L.nfl_grad_stash_bytes(C.byref(d), 8, 100, _lib.NFL_PREC_F16) == (8 * 4 + 1) * 173 * 1024 + 4096
    # split (hi + lo) records for the three-product backward: twice the record, the same mask words
    assert L.nfl_act_stash_bytes(C.byref(d), 8, 128, _lib.NFL_PREC_F16X3) == 8 * 4 * 2 * 178 * 1024 + 4096 + 8 * 4 * 84 * 256
    assert L.nfl_grad_stash_bytes(C.byref(d), 8, 100, _lib.NFL_PREC_F16X3) == (8 * 4 + 1) * 2 * 173 * 1024 + 4096
    assert L.nfl_act_stash_bytes(C.byref(d), 8, 128, _lib.NFL_PREC_F16W) == L.nfl_act_stash_bytes(C.byref(d), 8, 128, _lib.NFL_PREC_F16)
    assert L.nfl_grad_stash_bytes(C.byref(d), 8, 100, _lib.NFL_PREC_F16W) == L.nfl_grad_stash_bytes(C.byref(d), 8, 100, _lib.NFL_PREC_F16)


def test_encoder_widths(L):
    """--N_emb_xyz / --N_emb_dir are free integers in the reference (opt.py:25-28): 1..15 / 1..4 are built; a narrower
    encoder runs in the next wider kernel instantiation (4 or 6 k-steps of encoded position) on zero-padded weights."""
    import numpy as np
    for n_xyz, n_dir, nkp in ((10, 4, 4), (6, 2, 4), (3, 1, 4), (1, 1, 4), (11, 4, 6), (12, 3, 6), (15, 4, 6)):
        d = _lib.FieldDesc(n_xyz, n_dir, 1, 48, 1, 16, 0.1, 0)
        n = L.nfl_plan_bytes(C.byref(d))
        buf = C.create_string_buffer(n)
        assert L.nfl_plan_build(C.byref(d), _lib.NFL_PREC_F16X3, buf, n) == 0, (n_xyz, n_dir)
        hdr = np.frombuffer(buf.raw[:96], dtype=np.int32)
        assert hdr[6] == n_xyz and hdr[7] == nkp
        assert L.nfl_bwd_plan_build(C.byref(d), 1, _lib.NFL_PREC_F16, buf, n) == 0
        assert L.nfl_param_count(C.byref(d)) == sum(
            a * b + a for a, b in [(256, 6 * n_xyz + 3)] + [(256, 256)] * 3 + [(256, 256 + 6 * n_xyz + 3)] + [(256, 256)] * 4
            + [(128, 256 + 6 * n_dir + 3 + 48), (1, 256), (3, 128), (128, 256 + 16)] + [(128, 128)] * 3 + [(1, 128), (3, 128), (1, 128)])
        assert L.nfl_act_stash_bytes(C.byref(d), 2, 64, _lib.NFL_PREC_F16) == 2 * 2 * (nkp + 174) * 1024 + 4096 + 2 * 2 * 84 * 256
    # latent widths (opt.py --N_a / --N_tau): 1..48 / 1..16, zero-padded into the same k-steps
    for n_a, n_tau in ((24, 8), (40, 5), (1, 1), (33, 16)):
        d = _lib.FieldDesc(10, 4, 1, n_a, 1, n_tau, 0.1, 0)
        n = L.nfl_plan_bytes(C.byref(d))
        buf = C.create_string_buffer(n)
        assert L.nfl_plan_build(C.byref(d), _lib.NFL_PREC_F16X3, buf, n) == 0 and L.nfl_bwd_plan_build(C.byref(d), 0, _lib.NFL_PREC_F16, buf, n) == 0
    name = lambda n: L.nfl_render_kernel_name(_lib.NFL_PREC_F16X3, n).decode()
    assert name(6) == name(10) != name(12) == name(15)


def test_unsupported_configs_rejected(L):
    for bad in (_lib.FieldDesc(16, 4, 0, 48, 0, 16, 0.1, 0), _lib.FieldDesc(0, 4, 0, 48, 0, 16, 0.1, 0),
                _lib.FieldDesc(10, 5, 0, 48, 0, 16, 0.1, 0), _lib.FieldDesc(10, 4, 1, 49, 0, 16, 0.1, 0),
                _lib.FieldDesc(10, 4, 1, 48, 1, 17, 0.1, 0), _lib.FieldDesc(10, 4, 1, 0, 0, 16, 0.1, 0)):
        n = L.nfl_plan_bytes(C.byref(bad))
        buf = C.create_string_buffer(n)
        assert L.nfl_plan_build(C.byref(bad), 0, buf, n) == -1
    assert L.nfl_render_pass(None, None, None, None, None) == -1
    assert L.nfl_sample_pdf(None, None, None, None, 1, 64, 64, None, None, None) == -1


def test_render_rays_refuses_cpu_tensors():
    import torch
    from nerf_fl_amd import NeRF, PosEmbedding, render_rays
    models = {"coarse": NeRF("coarse")}
    emb = {"xyz": PosEmbedding(9, 10), "dir": PosEmbedding(3, 4)}
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        render_rays(models, emb, torch.zeros(4, 8), torch.zeros(4, dtype=torch.long), 8)


def test_auxiliary_entry_points_validate_arguments(L):
    """No GPU needed: the entry points around the render pass reject NULL / inconsistent arguments with NFL_EINVAL
    before touching the device."""
    assert L.nfl_adam_step(None, 1, 5e-4, 0.9, 0.999, 1e-8, 1, None) == -1
    t = _lib.AdamTensors()
    assert L.nfl_adam_step(C.byref(t), _lib.NFL_ADAM_MAX_TENSORS + 1, 5e-4, 0.9, 0.999, 1e-8, 1, None) == -1
    assert L.nfl_adam_step(C.byref(t), 1, 5e-4, 0.9, 0.999, 1e-8, 0, None) == -1          # steps are 1-based
    t.numel[0] = 4                                                                          # non-empty tensor without pointers
    assert L.nfl_adam_step(C.byref(t), 1, 5e-4, 0.9, 0.999, 1e-8, 1, None) == -1
    assert L.nfl_adam_step(C.byref(_lib.AdamTensors()), 0, 5e-4, 0.9, 0.999, 1e-8, 1, None) == 0
    assert L.nfl_loss_forward(None, None) == -1 and L.nfl_loss_backward(None, None) == -1
    a = _lib.LossArgs()
    a.n_rays = 4
    assert L.nfl_loss_forward(C.byref(a), None) == -1                                       # no tensors
    pose = (C.c_float * 12)()
    assert L.nfl_gen_rays(pose, 1.0, 1.0, 0.0, 0.0, 8, 0, 4, 2.0, 6.0, None, None) == -1   # no output buffer
    assert L.nfl_gen_rays(pose, 0.0, 1.0, 0.0, 0.0, 8, 0, 4, 2.0, 6.0, C.c_void_p(16), None) == -1   # fx = 0
    assert L.nfl_gen_rays(pose, 1.0, 1.0, 0.0, 0.0, 8, 0, 0, 2.0, 6.0, C.c_void_p(16), None) == 0    # empty range
    assert L.nfl_field_forward(None, None, None, None, 4, 90, 0, 0, None, None) == -1
    assert L.nfl_posenc(None, 4, 10, None, None, None) == -1
    assert L.nfl_composite_backward(None, None) == -1 and L.nfl_mlp_dgrad(None, None, None, None, None) == -1
    assert L.nfl_mlp_wgrad(None, None, None, None, None, 4, 64, 1, None, None, None, None) == -1
    # G (256 x 256) + the partial sums of 256 workgroups x 4 waves x (2 x 8 accumulator tiles x 64 lanes x 16 + 2 x 64 bias sums)
    assert L.nfl_wgrad_scratch_bytes() == (256 * 256 + 256 * 4 * (2 * 8 * 1024 + 2 * 64)) * 4
    assert L.nfl_pack_fields(0, None, None) == 0                                            # nothing to do
    assert L.nfl_pack_fields(1, None, None) == -1 and L.nfl_pack_fields(_lib.NFL_PACK_MAX_JOBS + 1, None, None) == -1
    assert L.nfl_compose_forward(None, 0, 0, 16, None, None, None, None, None) == -1        # no parameters
    fp = _lib.FieldParams()
    assert L.nfl_compose_forward(C.byref(fp), 0, 0, 16, C.c_void_p(16), C.c_void_p(16), None, None, None) == -1   # xyz_encoding_final missing
    job = (_lib.PackJob * 1)()                                                              # a job without a plan
    assert L.nfl_pack_fields(1, job, None) == -1


def test_adam_step_dev_validates_arguments(L):
    t = _lib.AdamTensors()
    assert L.nfl_adam_step_dev(None, 1, C.c_void_p(16), C.c_void_p(16), 1, None) == -1
    assert L.nfl_adam_step_dev(C.byref(t), 1, None, C.c_void_p(16), 1, None) == -1          # hyper-parameters missing
    assert L.nfl_adam_step_dev(C.byref(t), 1, C.c_void_p(16), None, 1, None) == -1          # step counter missing
    assert L.nfl_adam_step_dev(C.byref(t), 0, C.c_void_p(16), C.c_void_p(16), 1, None) == 0


def test_synth_generators_equal_the_oracles():
    """bench.py's GPU side draws its weights and rays from nerf_fl_amd.synth, its cpu_baseline leg and the tests from
    the oracle's own copies: the two must stay the same numbers."""
    import torch
    from nerf_fl_amd import synth
    from oracle import nerfw_oracle as orc
    for typ, kw in (("coarse", {}), ("fine", dict(encode_appearance=True, encode_transient=True)),
                    ("fine", dict(encode_appearance=True)), ("fine", dict(n_emb_xyz=15))):
        for regime in ("default", "sharp"):
            a = synth.make_field_params(7, regime, typ=typ, **kw)
            b = orc.make_field_params(orc.FieldSpec(typ, **kw), 7, regime)
            assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(synth.make_rays(100, 3, 0.5, 5.0), orc.make_rays(100, 3, 0.5, 5.0))
    assert torch.equal(synth.make_rays_photo(100, 3), orc.make_rays_photo(100, 3))


def test_product_package_never_imports_the_oracle():
    import glob
    for path in glob.glob(os.path.join(ROOT, "nerf_fl_amd", "*.py")):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path


def test_struct_layouts_match_the_header():
    """The ctypes mirrors of the argument structs must have the size the C compiler gives the header's structs (a field added on
    one side only -- e.g. nfl_dgrad_args.rounding_seed, ABI 9 -- would shift every later field silently)."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = {"nfl_field_desc": _lib.FieldDesc, "nfl_pass_args": _lib.PassArgs, "nfl_compbwd_args": _lib.CompBwdArgs,
             "nfl_dgrad_args": _lib.DgradArgs, "nfl_pack_job": _lib.PackJob, "nfl_camera": _lib.Camera,
             "nfl_field_params": _lib.FieldParams, "nfl_field_grads": _lib.FieldGrads, "nfl_adam_tensors": _lib.AdamTensors,
             "nfl_loss_args": _lib.LossArgs}
    src = '#include <stdio.h>\n#include "nerf_fl_amd.h"\nint main(void) {\n' + "".join(
        f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as td:
        c, exe = os.path.join(td, "sz.c"), os.path.join(td, "sz")
        with open(c, "w") as f:
            f.write(src)
        subprocess.run(["gcc", "-I", os.path.join(root, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n, cls in names.items():
        assert int(sizes[n]) == C.sizeof(cls), (n, sizes[n], C.sizeof(cls))


def test_rounding_seed_api():
    import nerf_fl_amd
    from nerf_fl_amd import rendering
    try:
        nerf_fl_amd.set_rounding_seed(2 ** 40 + 17)
        assert rendering._rounding_seed == 17                 # 32 bits travel through the C ABI
    finally:
        nerf_fl_amd.set_rounding_seed(0)
    assert rendering._rounding_seed == 0
