"""The C-ABI library loads (no GPU needed) and exports every symbol include/asd_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "asd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from scaledreamer_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/asd_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_host_side_grid_meta_matches_oracle(oracle):
    from scaledreamer_amd import _lib

    for args in [(16, 2, 19, 16, 1.447269237440378), (4, 2, 19, 4, 4.0), (16, 2, 19, 16, 1.0)]:
        a = _lib.make_grid_meta(*args)
        b = oracle.grid_meta(*args)
        assert bytes(a) == bytes(b)
    assert _lib.make_grid_meta(16, 2, 19, 16, 1.447269237440378).n_params == 12_599_920


def test_struct_layouts_match_between_binding_and_oracle(oracle):
    from scaledreamer_amd import _lib

    for a, b in [(_lib.GridMeta, oracle.GridMeta), (_lib.FieldCfg, oracle.FieldCfg), (_lib.MarchCfg, oracle.MarchCfg)]:
        assert ctypes.sizeof(a) == ctypes.sizeof(b)
        assert [(n, t) for n, t in a._fields_] == [(n, t) for n, t in b._fields_]


def test_errors_are_reported_not_swallowed():
    from scaledreamer_amd import _lib

    m = _lib.GridMeta()
    assert _lib.lib().asd_grid_meta_init(ctypes.byref(m), 17, 2, 19, 16, 1.5) == 0
    assert b"levels" in _lib.lib().asd_last_error()


def test_network_weight_tables_match_the_packers():
    """asd_unet_create / asd_vae_enc_create publish the packed weight table they read; weights.pack_unet / pack_vae_encoder must
    produce exactly those names and sizes (no device needed: handles own no device memory until weights are bound)."""
    import ctypes as C

    import torch

    from scaledreamer_amd._lib import WeightInfo, check, i32, lib
    from scaledreamer_amd.diffusion import weights as W
    from scaledreamer_amd.diffusion.engine import unet_desc
    from scaledreamer_amd.diffusion.vae_hip import vae_desc

    def table(kind, desc):
        h = C.c_void_p()
        check(getattr(lib(), f"asd_{kind}_create")(C.byref(desc), C.byref(h)))
        info, out = WeightInfo(), {}
        for i in range(getattr(lib(), f"asd_{kind}_num_weights")(h)):
            check(getattr(lib(), f"asd_{kind}_weight_info")(h, i32(i), C.byref(info)))
            out[info.name.decode()] = (info.rows, info.cols)
        assert getattr(lib(), f"asd_{kind}_weight_info")(h, i32(len(out)), C.byref(info)) != 0      # out of range -> error status
        getattr(lib(), f"asd_{kind}_destroy")(h)
        return out

    for cfg in (W.UNetConfig(), W.UNetConfig(camera_dim=16), W.UNetConfig(model_channels=64, context_dim=96, channel_mult=(1, 2), attention_resolutions=(1,))):
        packed = W.pack_unet({k: torch.empty(v, device="meta") for k, v in W.unet_layout(cfg)[0].items()}, cfg)
        t = table("unet", unet_desc(cfg))
        assert set(t) == set(packed)
        for k, (r, c) in t.items():
            assert packed[k].numel() == r * c, (k, tuple(packed[k].shape), r, c)
    t = table("unet", unet_desc(W.UNetConfig()))
    assert t["emb_all.weight"] == (sum(c for k, (r, c) in t.items() if k.endswith("in_layers.2.bias")), 1280)
    assert t["input_blocks.1.1.transformer_blocks.0.attn1.to_qk.weight"] == (640, 320) and t["input_blocks.0.0.weight"] == (320, 288)
    for cfg in (W.VAEConfig(), W.VAEConfig(ch=32)):
        packed = W.pack_vae_encoder({k: torch.zeros(v) for k, v in W.vae_encoder_layout(cfg)[0].items()}, cfg)
        t = table("vae_enc", vae_desc(cfg))
        assert set(t) == set(packed)
        for k, (r, c) in t.items():
            assert packed[k].numel() == r * c, (k, tuple(packed[k].shape), r, c)
    bad = unet_desc(W.UNetConfig())
    bad.num_head_channels = 32
    h = C.c_void_p()
    assert lib().asd_unet_create(C.byref(bad), C.byref(h)) != 0 and b"head_dim 64" in lib().asd_last_error()


def test_gemm_plan_table_round_trip():
    import ctypes as C

    from scaledreamer_amd._lib import GemmArgs, check, i32, lib

    n0 = lib().asd_gemm_plan_count()
    check(lib().asd_gemm_plan_set(i32(123456), i32(64), i32(72), i32(0), i32(72), i32(0), i32(0), i32(0), i32(0), i32(3), i32(2)))
    assert lib().asd_gemm_plan_count() == n0 + 1
    g = GemmArgs()
    g.M, g.N, g.K, g.lda = 123456, 64, 72, 72
    t, sk = C.c_int32(), C.c_int32()
    assert lib().asd_gemm_plan_get(C.byref(g), C.byref(t), C.byref(sk)) == 0 and (t.value, sk.value) == (3, 2)
    assert lib().asd_gemm_workspace_bytes(C.byref(g)) == 2 * 123456 * 64 * 4
    g.lda = 80                                   # another leading dimension is another shape: defaults, reported as un-tuned
    assert lib().asd_gemm_plan_get(C.byref(g), C.byref(t), C.byref(sk)) == 1 and t.value == 0 and sk.value >= 1
    assert lib().asd_gemm_plan_set(i32(1), i32(1), i32(1), i32(0), i32(0), i32(0), i32(0), i32(0), i32(0), i32(99), i32(1)) != 0
