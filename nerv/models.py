"""nerv.models: the conv / deconv block builders the SAVi encoder and decoder are assembled from (savi.py:231-275)."""
from slotformer_amd.nerv_compat import conv_norm_act, deconv_norm_act, deconv_out_shape  # noqa: F401

__all__ = ['conv_norm_act', 'deconv_norm_act', 'deconv_out_shape']
