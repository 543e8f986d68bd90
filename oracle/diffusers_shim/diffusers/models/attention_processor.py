"""diffusers.models.attention_processor shim: Attention (module container).

The arithmetic lives in the *processor* (the reference's own
AttentionProcessor); this class only owns the parameters and forwards kwargs
that the processor's __call__ accepts, as diffusers does.
"""
import inspect

import torch.nn as nn

from .normalization import RMSNorm


class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim=None, heads: int = 8,
                 kv_heads=None, dim_head: int = 64, dropout: float = 0.0,
                 bias: bool = False, upcast_attention: bool = False,
                 upcast_softmax: bool = False, cross_attention_norm=None,
                 cross_attention_norm_num_groups: int = 32, qk_norm=None,
                 added_kv_proj_dim=None, norm_num_groups=None, spatial_norm_dim=None,
                 out_bias: bool = True, scale_qk: bool = True,
                 only_cross_attention: bool = False, eps: float = 1e-5,
                 rescale_output_factor: float = 1.0, residual_connection: bool = False,
                 _from_deprecated_attn_block: bool = False, processor=None,
                 out_dim=None, out_context_dim=None, context_pre_only=None,
                 pre_only=False, elementwise_affine: bool = True, is_causal: bool = False):
        super().__init__()
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.inner_kv_dim = self.inner_dim if kv_heads is None else dim_head * kv_heads
        self.query_dim = query_dim
        self.use_bias = bias
        self.is_cross_attention = cross_attention_dim is not None
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.dropout = dropout
        self.out_dim = out_dim if out_dim is not None else query_dim
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = out_dim // dim_head if out_dim is not None else heads
        self.group_norm = None
        self.spatial_norm = None

        if qk_norm is None:
            self.norm_q = None
            self.norm_k = None
        elif qk_norm == "rms_norm":
            self.norm_q = RMSNorm(dim_head, eps=eps)
            self.norm_k = RMSNorm(dim_head, eps=eps)
        else:
            raise NotImplementedError(f"shim: qk_norm={qk_norm}")

        if cross_attention_norm is None:
            self.norm_cross = None
        elif cross_attention_norm == "layer_norm":
            self.norm_cross = nn.LayerNorm(self.cross_attention_dim)
        else:
            raise NotImplementedError(f"shim: cross_attention_norm={cross_attention_norm}")

        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_kv_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_kv_dim, bias=bias)
        self.to_out = nn.ModuleList([])
        self.to_out.append(nn.Linear(self.inner_dim, self.out_dim, bias=out_bias))
        self.to_out.append(nn.Dropout(dropout))
        self.processor = processor

    def norm_encoder_hidden_states(self, encoder_hidden_states):
        assert self.norm_cross is not None
        return self.norm_cross(encoder_hidden_states)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None,
                **cross_attention_kwargs):
        attn_parameters = set(inspect.signature(self.processor.__call__).parameters.keys())
        cross_attention_kwargs = {
            k: w for k, w in cross_attention_kwargs.items() if k in attn_parameters
        }
        return self.processor(
            self,
            hidden_states,
            encoder_hidden_states=encoder_hidden_states,
            attention_mask=attention_mask,
            **cross_attention_kwargs,
        )
