"""ctypes binding of libvp_hip.so (include/vp_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VP_LIB", os.path.join(_HERE, "libvp_hip.so"))   # VP_LIB: ablation builds only

VP_F32, VP_BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


class PixReferDesc(ctypes.Structure):
  _fields_ = [("batch", ctypes.c_int), ("height", ctypes.c_int), ("ngf", ctypes.c_int), ("ndf", ctypes.c_int),
              ("dtype", ctypes.c_int), ("training", ctypes.c_int), ("l1_weight", ctypes.c_float),
              ("gan_weight", ctypes.c_float), ("per_sample_bn", ctypes.c_int),
              # schedule of a training plan (0 = default; include/vp_hip.h)
              ("streams", ctypes.c_int), ("d_backward_fork", ctypes.c_int), ("d_beside_vgg", ctypes.c_int)]


class ConvDesc(ctypes.Structure):
  _fields_ = [("kind", ctypes.c_int), ("n", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int),
              ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("ksize", ctypes.c_int), ("stride", ctypes.c_int),
              ("pad", ctypes.c_int), ("dtype", ctypes.c_int), ("in_act", ctypes.c_int), ("out_act", ctypes.c_int)]


class LogMelDesc(ctypes.Structure):
  _fields_ = [("sample_rate", ctypes.c_int), ("num_mel_bins", ctypes.c_int), ("win_length", ctypes.c_int),
              ("hop_step", ctypes.c_int), ("fft_length", ctypes.c_int), ("lower_hz", ctypes.c_float),
              ("upper_hz", ctypes.c_float), ("batch", ctypes.c_int), ("samples", ctypes.c_int)]


class BfmNetDesc(ctypes.Structure):
  _fields_ = [("batch", ctypes.c_int), ("frames", ctypes.c_int), ("num_mel_bins", ctypes.c_int), ("trunk_dtype", ctypes.c_int)]


class BfmModel(ctypes.Structure):
  _fields_ = [("nver", ctypes.c_int), ("ntri", ctypes.c_int), ("meanshape", ctypes.c_void_p), ("idBase", ctypes.c_void_p),
              ("exBase", ctypes.c_void_p), ("meantex", ctypes.c_void_p), ("texBase", ctypes.c_void_p), ("tri", ctypes.c_void_p),
              ("point_buf", ctypes.c_void_p), ("center", ctypes.c_double * 3), ("focal", ctypes.c_double),
              ("image_center", ctypes.c_double), ("sh", ctypes.c_double * 5)]


_P = ctypes.c_void_p
_SIGNATURES = {
    "vp_version": (ctypes.c_int, []),
    "vp_last_error": (ctypes.c_char_p, []),
    "vp_pixrefer_param_count": (ctypes.c_size_t, [ctypes.POINTER(PixReferDesc), ctypes.c_int]),
    "vp_pixrefer_param_info": (ctypes.c_int, [ctypes.POINTER(PixReferDesc), ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                              ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int),
                                              ctypes.POINTER(ctypes.c_int64)]),
    "vp_pixrefer_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(PixReferDesc)]),
    "vp_pixrefer_validate_plan": (ctypes.c_int, [ctypes.POINTER(PixReferDesc)]),
    "vp_pixrefer_create": (ctypes.c_int, [ctypes.POINTER(PixReferDesc), _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P,
                                          ctypes.POINTER(_P)]),
    "vp_pixrefer_destroy": (None, [_P]),
    "vp_pixrefer_params_changed": (ctypes.c_int, [_P]),
    "vp_pixrefer_optimizer_stepped": (ctypes.c_int, [_P]),
    "vp_pixrefer_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    "vp_pixrefer_forward_fg3": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "vp_pixrefer_desc_size": (ctypes.c_size_t, []),
    "vp_pixrefer_backward": (ctypes.c_int, [_P, _P]),
    "vp_pixrefer_backward_update": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                   ctypes.c_float, ctypes.c_float, _P]),
    "vp_pixrefer_backward_d": (ctypes.c_int, [_P, _P]),
    "vp_pixrefer_backward_g": (ctypes.c_int, [_P, _P]),
    "vp_pixrefer_backward_d_fork": (ctypes.c_int, [_P, _P]),
    "vp_pixrefer_backward_d_join": (ctypes.c_int, [_P, _P]),
    "vp_pixrefer_backward_g_stages": (ctypes.c_int, []),
    "vp_pixrefer_backward_g_stage": (ctypes.c_int, [_P, ctypes.c_int, _P]),
    "vp_pixrefer_side_stream": (ctypes.c_void_p, [_P]),
    "vp_pixrefer_use_streams": (ctypes.c_int, [_P, ctypes.c_int]),
    "vp_pixrefer_set_option": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int]),
    "vp_pixrefer_fetch": (ctypes.c_int, [_P, ctypes.c_int, _P, _P]),
    "vp_pixrefer_counter": (ctypes.c_longlong, [_P, ctypes.c_char_p]),
    "vp_crc32c": (ctypes.c_uint, [_P, ctypes.c_size_t, ctypes.c_uint]),
    "vp_pixrefer_phase_ms": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
    "vp_grad_pack_bf16": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P]),
    "vp_grad_unpack_bf16": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_float, _P]),
    "vp_pixrefer_update_bucket": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                 ctypes.c_float, _P]),
    "vp_resize_paste_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "vp_resize_paste_u8": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_resize_linear_table": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P]),
    "vp_mm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vp_mm_fwd_f32": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_mm_bwd_data_f32": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, _P, _P]),
    "vp_mm_bwd_weight_f32": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_mm_packed_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vp_mm_pack_desc_bytes": (ctypes.c_size_t, []),
    "vp_mm_pack_desc": (ctypes.c_int, [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, _P]),
    "vp_mm_pack_table": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _P]),
    "vp_mm_fwd_f32_packed": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_mm_bwd_data_f32_packed": (ctypes.c_int, [_P, ctypes.c_int, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_pixrefer_mark_ms": (ctypes.c_float, [_P, ctypes.c_int, ctypes.c_int]),
    "vp_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "vp_tune": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "vp_reserve_streams": (ctypes.c_int, []),
    "vp_host_stream": (ctypes.c_void_p, []),
    "vp_pixrefer_pack_frames": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, _P]),
    "vp_profile_collect": (ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t]),
    "vp_pixrefer_tensor": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_int)]),
    "vp_adam_tf": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_float, _P]),
    "vp_conv_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "vp_conv_fwd": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "vp_conv_bwd_data": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "vp_conv_bwd_weight": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "vp_bn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vp_bn_stats": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P]),
    "vp_logmel_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(LogMelDesc)]),
    "vp_logmel_frames": (ctypes.c_int, [ctypes.POINTER(LogMelDesc)]),
    "vp_logmel_create": (ctypes.c_int, [ctypes.POINTER(LogMelDesc), _P, ctypes.c_size_t, _P, ctypes.POINTER(_P)]),
    "vp_logmel_destroy": (None, [_P]),
    "vp_logmel_forward": (ctypes.c_int, [_P, _P, _P, _P]),
    "vp_bfmnet_param_count": (ctypes.c_size_t, []),
    "vp_bfmnet_param_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]),
    "vp_bfmnet_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(BfmNetDesc)]),
    "vp_bfmnet_create": (ctypes.c_int, [ctypes.POINTER(BfmNetDesc), _P, ctypes.c_size_t, _P, _P, ctypes.POINTER(_P)]),
    "vp_bfmnet_destroy": (None, [_P]),
    "vp_bfmnet_params_changed": (ctypes.c_int, [_P]),
    "vp_bfmnet_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    "vp_bfmnet_set_decoder_dropout": (ctypes.c_int, [_P, _P, _P]),
    "vp_bfmnet_tensor": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    "vp_maxpool2x2_fwd": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "vp_maxpool2x2_bwd": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "vp_composite_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    "vp_gan_loss": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_float, ctypes.c_int, _P]),
    "vp_dwconv7x3_bn_act": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "vp_maxpool_hw": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, _P]),
    "vp_gru_seq": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    "vp_bfm_reconstruct_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vp_bfm_reconstruct": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "vp_render_colors_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vp_render_colors": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _P, _P]),
    "vp_bn_train_workspace_bytes": (ctypes.c_size_t, [ctypes.c_size_t, ctypes.c_int]),
    "vp_bn_train_fwd": (ctypes.c_int, [_P, ctypes.c_size_t, ctypes.c_int, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "vp_bn_train_bwd": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_int, _P, _P, _P, _P, _P, _P]),
    "vp_affine_act_fwd": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_affine_act_add_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_act_bwd": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, ctypes.c_int, _P, _P]),
    "vp_dwconv7x3_raw": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "vp_dwconv7x3_bwd_data": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "vp_dwconv7x3_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vp_bn_act_train_bwd": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_int, _P, _P, _P, ctypes.c_int, _P, _P, _P, _P]),
    "vp_l2_regulariser": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, ctypes.c_float, _P, _P]),
    "vp_adam_tf_clipped": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, _P, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P]),
    "vp_moving_update": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, ctypes.c_float, _P]),
    "vp_dwconv7x3_wgrad": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_maxpool_hw_bwd": (ctypes.c_int, [_P, _P, _P] + [ctypes.c_int] * 8 + [_P]),
    "vp_stem_im2col": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "vp_gru_train_fwd": (ctypes.c_int, [_P] * 10 + [ctypes.c_int, ctypes.c_int, _P]),
    "vp_gru_train_bwd": (ctypes.c_int, [_P] * 10 + [ctypes.c_int, ctypes.c_int, _P]),
    "vp_colsum_f32": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, _P]),
    "vp_gru_split_recurrent": (ctypes.c_int, [_P] * 7),
    "vp_mul_f32": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, _P]),
    "vp_add_ears_f32": (ctypes.c_int, [_P, _P, ctypes.c_int, _P]),
    "vp_vertex_loss_partials": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "vp_bfm_vertex_loss": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "vp_sumsq_partials": (ctypes.c_int, [ctypes.c_size_t]),
    "vp_sum_f64": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_double, _P, _P, _P]),
    "vp_bfm_step_report": (ctypes.c_int, [_P, _P, ctypes.c_double, _P, _P, _P]),
    "vp_clip_scale_f32": (ctypes.c_int, [_P, ctypes.c_size_t, _P, ctypes.c_float, _P]),
    "vp_sumsq": (ctypes.c_int, [_P, ctypes.c_size_t, _P, _P]),
    "vp_bn_bwd": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def lib():
  """The loaded library; raises (never falls back) when it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError("libvp_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C voicepuppet_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # PyTorch first: it carries its own HIP runtime (torch/lib/libamdhip64.so); a process that loaded the system runtime through THIS
    # library before importing torch ends up with two runtimes, and the second finds no device ("no ROCm-capable device is detected" from
    # the first HIP call of the library - seen with __graft_entry__.build() followed by smoke() in one process).  Loaded in this order
    # the library binds to the runtime torch already mapped.
    try:
      import torch  # noqa: F401
    except ImportError:
      pass
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
      try:
        fn = getattr(l, name)
      except AttributeError:
        if "VP_LIB" not in os.environ:       # the shipped library exports everything include/vp_hip.h declares (tests/test_host_logic.py)
          raise
        continue                             # an A/B build of an older tree (scripts/ab.sh): entry points added since are simply absent
      fn.restype = res
      fn.argtypes = args
    # the descriptor is declared twice (include/vp_hip.h, PixReferDesc above): a library built from another header must not be handed
    # this layout (include/vp_hip.h, "ABI rule"); an older A/B build (VP_LIB) without the query is the 48-byte round-5 layout
    if hasattr(l, "vp_pixrefer_desc_size") and l.vp_pixrefer_desc_size.argtypes is not None:
      want = int(l.vp_pixrefer_desc_size())
      if want != ctypes.sizeof(PixReferDesc):
        raise RuntimeError("%s: vp_pixrefer_desc is %d bytes in the library, %d in this binding" % (LIB_PATH, want, ctypes.sizeof(PixReferDesc)))
    _lib = l
  return _lib


def exported_symbols():
  return sorted(_SIGNATURES)


def check(rc, what=""):
  if rc != 0:
    raise RuntimeError("%s failed (%d): %s" % (what or "libvp_hip call", rc, lib().vp_last_error().decode()))
