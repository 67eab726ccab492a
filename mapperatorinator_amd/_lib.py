"""ctypes binding of libmapperhip.so (the C ABI declared in include/mapperhip.h).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU fallback and
nothing here imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os

MH_MAX_LAYERS = 32
MH_F32, MH_BF16, MH_MX8 = 0, 1, 2
(EPI_STORE, EPI_STORE_F32, EPI_RESID, EPI_GEGLU, EPI_BIAS_GELU, EPI_GATE_RESID, EPI_KV_SCATTER,
 EPI_QKV_VT, EPI_QKV_CACHE, EPI_BIAS_GELU_ERF) = range(10)

# MAPPERHIP_LIB: developer override (tools/decode_phases.py loads the profiling build); the package default is the
# in-tree production library
_LIB_PATH = os.environ.get("MAPPERHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmapperhip.so")

VP = C.c_void_p
FP = C.c_void_p  # device float*
_PTR_ARR = VP * MH_MAX_LAYERS


class MhGemm(C.Structure):
    _fields_ = [("A", VP), ("lda", C.c_int), ("W", VP), ("ldw", C.c_int), ("C", VP), ("ldc", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("bias", VP), ("gate", VP),
                ("gate_ld", C.c_int), ("rows_per_batch", C.c_int), ("kv_B", C.c_int), ("kv_H", C.c_int),
                ("kv_L", C.c_int), ("C2", VP), ("n_split", C.c_int), ("kv_Lpad", C.c_int),
                ("C3", VP), ("C4", VP), ("cache_len", C.c_int),
                ("dtype", C.c_int), ("epilogue", C.c_int),
                ("stats_out", VP), ("ln_stats", VP), ("ln_strips", C.c_int), ("ln_shift", VP), ("ln_scale", VP),
                ("ln_ld", C.c_int), ("ln_eps", C.c_float), ("w_split3", C.c_int),
                ("a_scale", VP), ("w_scale", VP),      # ABI 7: MH_MX8 operands
                ("mx_out", VP), ("mx_out_scales", VP)]   # ABI 9: MX-fp8 image of a GEGLU / BIAS_GELU result


class MhT5Config(C.Structure):
    _fields_ = [("d_model", C.c_int), ("d_kv", C.c_int), ("d_ff", C.c_int), ("n_heads", C.c_int),
                ("n_enc_layers", C.c_int), ("n_dec_layers", C.c_int), ("vocab_in", C.c_int),
                ("vocab_out", C.c_int), ("n_mels", C.c_int), ("n_mels_pad", C.c_int), ("src_len", C.c_int),
                ("tgt_len", C.c_int), ("dtype", C.c_int), ("eps", C.c_float),
                # ABI 5: the Whisper-family backbone (arch 1)
                ("arch", C.c_int), ("attn_scale", C.c_float), ("in_frames", C.c_int), ("local_every", C.c_int),
                ("local_window", C.c_int),
                ("enc_operand_dtype", C.c_int),      # ABI 7: 0 or MH_MX8
                ("options", VP),                     # ABI 8: MhOptionSet* of this engine (NULL = process-wide values)
                ("dec_pos_from_mask", C.c_int)]      # ABI 10 (arch 2)


class MhT5Weights(C.Structure):
    _fields_ = [("enc_embed_w", VP), ("enc_embed_b", VP), ("dec_embed", VP), ("enc_rel_bias", VP),
                ("dec_rel_bias", VP),
                ("enc_ln1", _PTR_ARR), ("enc_qkv", _PTR_ARR), ("enc_o", _PTR_ARR), ("enc_ln2", _PTR_ARR),
                ("enc_wi", _PTR_ARR), ("enc_wo", _PTR_ARR), ("enc_final_ln", VP),
                ("dec_ln1", _PTR_ARR), ("dec_qkv", _PTR_ARR), ("dec_o", _PTR_ARR), ("dec_ln2", _PTR_ARR),
                ("dec_cq", _PTR_ARR), ("dec_ckv_all", VP), ("dec_co", _PTR_ARR), ("dec_ln3", _PTR_ARR),
                ("dec_wi", _PTR_ARR), ("dec_wo", _PTR_ARR), ("dec_final_ln", VP), ("lm_head", VP),
                # ABI 5, arch 1 only
                ("conv1_w", VP), ("conv1_b", VP), ("conv2_w", VP), ("conv2_b", VP),
                ("enc_qkv_b", _PTR_ARR), ("enc_o_b", _PTR_ARR), ("enc_fc1_b", _PTR_ARR), ("enc_fc2_b", _PTR_ARR),
                ("dec_qkv_b", _PTR_ARR), ("dec_o_b", _PTR_ARR), ("dec_cq_b", _PTR_ARR), ("dec_ckv_b_all", VP),
                ("dec_co_b", _PTR_ARR), ("dec_fc1_b", _PTR_ARR), ("dec_fc2_b", _PTR_ARR),
                ("enc_rope", VP), ("enc_rope_local", VP), ("dec_rope", VP), ("dec_rope_local", VP),
                # ABI 7: MX-fp8 copies of the encoder projections and of the cross-K/V projection
                ("enc_qkv_mx", _PTR_ARR), ("enc_qkv_mxs", _PTR_ARR), ("enc_o_mx", _PTR_ARR), ("enc_o_mxs", _PTR_ARR),
                ("enc_wi_mx", _PTR_ARR), ("enc_wi_mxs", _PTR_ARR), ("enc_wo_mx", _PTR_ARR), ("enc_wo_mxs", _PTR_ARR),
                ("dec_ckv_all_mx", VP), ("dec_ckv_all_mxs", VP),
                # ABI 10, arch 2 (HF Whisper): LayerNorm biases, absolute position tables
                ("enc_ln1_b", _PTR_ARR), ("enc_ln2_b", _PTR_ARR), ("enc_final_ln_b", VP),
                ("dec_ln1_b", _PTR_ARR), ("dec_ln2_b", _PTR_ARR), ("dec_ln3_b", _PTR_ARR), ("dec_final_ln_b", VP),
                ("enc_pos", VP), ("dec_pos", VP)]


class MhSampling(C.Structure):
    _fields_ = [("do_sample", C.c_int), ("top_k", C.c_int), ("top_p", C.c_float), ("temperature", C.c_float),
                ("timeshift_bias", C.c_float), ("ts_start", C.c_int), ("ts_end", C.c_int), ("n_sos", C.c_int),
                ("sos_ids", C.c_int * 16), ("lookback_mask_end", C.c_int), ("pad_id", C.c_int),
                ("max_length", C.c_int), ("seed", C.c_uint64),
                ("cfg_scale", C.c_float), ("n_cond", C.c_int), ("cond_temp", C.c_float * 3),
                ("cond_offset", C.c_int * 3), ("lookback_types_first", C.c_int), ("tok_flags", VP),
                ("cond_per_row", C.c_int), ("rng_row0", C.c_uint),
                ("cross_kv_fp8", C.c_void_p)]


class MhBeamStep(C.Structure):
    _fields_ = [("logits", VP), ("eos_table", VP),
                ("G", C.c_int), ("num_beams", C.c_int), ("V", C.c_int), ("P", C.c_int), ("max_length", C.c_int), ("K", C.c_int),
                ("cur_len", C.c_int), ("cfg", C.c_int), ("cfg_scale", C.c_float), ("length_penalty", C.c_float),
                ("early_stopping", C.c_int), ("sp", MhSampling),
                ("run_in", VP), ("rs_in", VP), ("rb_in", VP), ("seq_in", VP), ("bs_in", VP), ("bb_in", VP), ("fin_in", VP),
                ("run_out", VP), ("rs_out", VP), ("rb_out", VP), ("seq_out", VP), ("bs_out", VP), ("bb_out", VP), ("fin_out", VP),
                ("heuristic_open", VP), ("src", VP), ("last", VP), ("flags", VP)]


class MhDiTConfig(C.Structure):
    _fields_ = [("hidden", C.c_int), ("depth", C.c_int), ("n_heads", C.c_int), ("context_size", C.c_int),
                ("class_size", C.c_int), ("in_channels", C.c_int), ("freq_dim", C.c_int),
                ("t_freq_dim", C.c_int), ("first_k_pad", C.c_int), ("class_pad", C.c_int), ("operand_dtype", C.c_int),
                ("options", VP)]      # ABI 8


class MhDiTWeights(C.Structure):
    _fields_ = [("pos_freqs", VP), ("t_freqs", VP), ("first_w", VP), ("first_b", VP),
                ("t_w0", VP), ("t_b0", VP), ("t_w1", VP), ("t_b1", VP),
                ("y_w0", VP), ("y_b0", VP), ("y_w1", VP), ("y_b1", VP),
                ("ada_w", _PTR_ARR), ("ada_b", _PTR_ARR), ("qkv_w", _PTR_ARR), ("qkv_b", _PTR_ARR),
                ("out_w", _PTR_ARR), ("out_b", _PTR_ARR), ("fc1_w", _PTR_ARR), ("fc1_b", _PTR_ARR),
                ("fc2_w", _PTR_ARR), ("fc2_b", _PTR_ARR), ("fin_ada_w", VP), ("fin_ada_b", VP),
                ("fin_w", VP), ("fin_b", VP),
                ("first_w3", VP), ("qkv_w3", _PTR_ARR), ("out_w3", _PTR_ARR), ("fc1_w3", _PTR_ARR), ("fc2_w3", _PTR_ARR),
                ("qkv_wb", _PTR_ARR), ("out_wb", _PTR_ARR), ("fc1_wb", _PTR_ARR), ("fc2_wb", _PTR_ARR),
                # ABI 7: MX-fp8 copies (elements, scales) of the block projections
                ("qkv_wm", _PTR_ARR), ("qkv_wms", _PTR_ARR), ("out_wm", _PTR_ARR), ("out_wms", _PTR_ARR),
                ("fc1_wm", _PTR_ARR), ("fc1_wms", _PTR_ARR), ("fc2_wm", _PTR_ARR), ("fc2_wms", _PTR_ARR)]


class MhSliderSet(C.Structure):
    _fields_ = [("n_chunks", C.c_int32), ("pair_stride", C.c_int32), ("n_sliders", C.c_int32),
                ("chunk_active", VP), ("chunk_off", VP), ("type", VP), ("cp_off", VP), ("cp_idx", VP),
                ("end_idx", VP), ("length", VP)]


ABI_VERSION = 10  # MH_ABI_VERSION of include/mapperhip.h

# every symbol include/mapperhip.h declares: (name, restype, argtypes)
I, I64, F = C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "mh_last_error": (C.c_char_p, []),
    "mh_abi_version": (I, []),
    "mh_struct_size": (I, [I]),
    "mh_set_option": (I, [C.c_char_p, C.c_long]),
    "mh_get_option": (C.c_long, [C.c_char_p]),
    "mh_options_create": (VP, []),
    "mh_options_destroy": (None, [VP]),
    "mh_options_set": (I, [VP, C.c_char_p, C.c_long]),
    "mh_options_clear": (I, [VP, C.c_char_p]),
    "mh_options_get": (C.c_long, [VP, C.c_char_p]),
    "mh_mel": (I, [VP, I, I, I, I, I, VP, VP, VP, VP, VP, VP, I, VP, I, I, VP]),
    "mh_gemm": (I, [C.POINTER(MhGemm), VP]),
    "mh_rmsnorm": (I, [VP, I, VP, VP, I, I, I, F, I, VP]),
    "mh_layernorm": (I, [VP, I, VP, VP, VP, I, I, I, F, I, VP]),
    "mh_mx8_scale_row_bytes": (I64, [I]),
    "mh_quantize_mx8": (I, [VP, I, I, I, I, VP, I, VP, VP]),
    "mh_rmsnorm_mx8": (I, [VP, I, VP, I, I, F, I, VP, I, VP, VP]),
    "mh_attention": (I, [VP, I, I, VP, I, VP, VP, I, I, I, I, F, I, I, VP]),
    "mh_whisper_frontend_workspace_bytes": (I64, [I, I, I, I, I]),
    "mh_whisper_frontend": (I, [VP, I, I, I, VP, VP, VP, VP, VP, I, VP, VP, I64, I, VP]),
    "mh_cond_channels": (I, [VP, I, I, I, I, VP, I, I, VP]),
    "mh_t5_encode_workspace_bytes": (I64, [C.POINTER(MhT5Config), I]),
    "mh_t5_encode": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, VP, VP, VP, I64, VP]),
    "mh_t5_cross_kv": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, VP, VP]),
    "mh_t5_cross_kv_workspace_bytes": (I64, [C.POINTER(MhT5Config), I]),
    "mh_t5_cross_kv_ws": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, VP, VP, I64, VP]),
    "mh_t5_encode_cond": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, VP, VP, VP, VP, I64, VP]),
    "mh_t5_decode_workspace_bytes": (I64, [C.POINTER(MhT5Config), I]),
    "mh_t5_cross_kv_fp8_bytes": (I64, [C.POINTER(MhT5Config), I]),
    "mh_t5_quantize_cross_kv": (I, [C.POINTER(MhT5Config), VP, I, VP, VP]),
    "mh_t5_generate": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, VP, VP, I, VP,
                           C.POINTER(MhSampling), VP, VP, VP, VP, VP, I64, I, VP]),
    "mh_t5_step": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, I, VP, I, VP, I, VP, VP, I64, VP]),
    "mh_beam_step_lds_bytes": (I64, [I, I]),
    "mh_beam_step": (I, [C.POINTER(MhBeamStep), VP]),
    "mh_t5_reorder_cache_scratch_bytes": (I64, [C.POINTER(MhT5Config), I, I]),
    "mh_t5_reorder_cache": (I, [C.POINTER(MhT5Config), I, VP, I, VP, I64, VP, I64, VP]),
    "mh_t5_forward_workspace_bytes": (I64, [C.POINTER(MhT5Config), I, I]),
    "mh_t5_decoder_forward": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, VP, VP, I, VP, VP, I64, VP]),
    "mh_t5_cross_attn_probe": (I, [C.POINTER(MhT5Config), C.POINTER(MhT5Weights), VP, I, I, C.POINTER(C.c_float), VP, I64, VP]),
    "mh_t5_decode_timing": (I, [VP, I]),
    "mh_t5_decode_chains": (I, [I]),
    "mh_t5_decode_chains_cfg": (I, [C.POINTER(MhT5Config), I]),
    "mh_t5_step_graph_cache_stats": (I, [C.POINTER(C.c_long), C.POINTER(C.c_long), I]),
    "mh_wall_clock_khz": (I, []),
    "mh_dit_workspace_bytes": (I64, [C.POINTER(MhDiTConfig), I, I]),
    "mh_dit_forward_cfg": (I, [C.POINTER(MhDiTConfig), C.POINTER(MhDiTWeights), VP, VP, VP, VP, F, I, I, I, I,
                               VP, VP, I64, VP]),
    "mh_ddpm_step": (I, [VP, VP, VP, VP, VP, VP, VP, I, I, I, VP, VP, VP]),
    "mh_ddpm_loop_workspace_bytes": (I64, [C.POINTER(MhDiTConfig), I, I, I]),
    "mh_ddpm_sample_loop": (I, [C.POINTER(MhDiTConfig), C.POINTER(MhDiTWeights), VP, VP, VP, F, I, I, I, I, I,
                                VP, VP, VP, VP, VP, C.POINTER(MhSliderSet), VP, I64, VP]),
    "mh_slider_project": (I, [VP, VP, VP, I, I, C.POINTER(MhSliderSet), VP]),
}

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libmapperhip.so once; raise RuntimeError (never fall back) if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"libmapperhip.so not found at {_LIB_PATH}: build it with `make` (or "
            f"`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libmapperhip.so does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    if lib.mh_abi_version() != ABI_VERSION:
        raise RuntimeError("libmapperhip.so ABI version mismatch")
    for which, st in enumerate((MhGemm, MhT5Config, MhT5Weights, MhSampling, MhDiTConfig, MhDiTWeights, MhSliderSet, MhBeamStep)):
        if lib.mh_struct_size(which) != C.sizeof(st):
            raise RuntimeError(f"libmapperhip.so: layout of {st.__name__} differs from the binding "
                               f"({lib.mh_struct_size(which)} vs {C.sizeof(st)} bytes)")
    _lib = lib
    return lib


def set_option(name: str, value: int) -> int:
    """mh_set_option: returns the previous value (so that tests can restore it)."""
    lib = load()
    old = lib.mh_get_option(name.encode())
    check(lib.mh_set_option(name.encode(), int(value)), f"mh_set_option({name})")
    return old


class OptionSet:
    """An engine's own option overrides (MhOptionSet, ABI 8): `T5Engine(..., options={"decode_chains": 1})` runs with them
    while another engine in the same process keeps the process-wide values.  `handle` goes into MhT5Config / MhDiTConfig
    `.options`; the object must outlive the engine's calls (the engines keep a reference)."""

    def __init__(self, values: dict = None):
        self._lib = load()
        self.handle = self._lib.mh_options_create()
        if not self.handle:
            raise MemoryError("mh_options_create")
        for k, v in (values or {}).items():
            self[k] = v

    def __setitem__(self, name: str, value: int):
        check(self._lib.mh_options_set(self.handle, name.encode(), int(value)), f"mh_options_set({name})")

    def __getitem__(self, name: str) -> int:
        return int(self._lib.mh_options_get(self.handle, name.encode()))

    def clear(self, name: str = None):
        check(self._lib.mh_options_clear(self.handle, None if name is None else name.encode()), "mh_options_clear")

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self._lib.mh_options_destroy(h)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mh_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libmapperhip {what} failed (status {rc}): {msg}")


def ptr(t) -> int:
    """device pointer of a torch tensor (or None -> NULL)"""
    if t is None:
        return None
    return t.data_ptr()
