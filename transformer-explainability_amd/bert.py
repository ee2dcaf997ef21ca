"""BERT (sequence classification) with relevance propagation on MI355X kernels.

Host-side mirror of BERT_explainability/modules/BERT/BERT.py + BertForSequenceClassification.py
(variant "ours") and BERT_orig_lrp.py + BERT_cls_lrp.py (variant "lrp") of the reference: same module
tree and parameter names as Hugging Face BERT (so ``bert-base-uncased`` style state dicts load), same
accessors (``layer.attention.self.get_attn()/get_attn_cam()/get_attn_gradients()``) and
``model.relprop(one_hot, alpha=1)``.  The classes are plain ``nn.Module``s configured from any object
with the usual BertConfig attributes -- the reference subclasses transformers' BertPreTrainedModel,
whose 3.5.1 API no longer exists in current transformers.

forward/backward: stock PyTorch-ROCm.  relprop: HIP kernels; q/k/v relevance goes from the attention
rules to the three Linear rules without head-transpose copies; the /2 of BERT.py:373-374,392-393 is
folded into the kernels' store.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from . import rules as R_ours

__all__ = ["BertConfigLite", "make_bert_module", "BertModel", "BertForSequenceClassification"]


def BertConfigLite(**kw):
    """Minimal stand-in for transformers.BertConfig (bert-base-uncased defaults)."""
    d = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
             intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
             attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
             layer_norm_eps=1e-12, pad_token_id=0, num_labels=2)
    d.update(kw)
    return SimpleNamespace(**d)


class _FusedSelfAttention(torch.autograd.Function):
    """the core of BertSelfAttention.forward (BERT.py:336-352: scores, / sqrt(D), + mask, softmax,
    probs v) on the producer kernels of csrc/te_attn_long.hip instead of ~10 stock launches per direction.

    forward : q, k, v [B,N,C] (the three Linear outputs, read in place as [B,H,N,D] views) -> ctx [B,N,C]; the
              probabilities, the unscaled scores (matmul1.Y) and the masked scaled scores (add.X[0]) come back as
              non-differentiable by-products -- the tensors the relprop rules read.
    backward: d_attn -- the attention gradient the explanation needs -- goes to the module (save_attn_gradients, what
              the reference's register_hook does, BERT.py:347-348); d_q / d_k / d_v for the layers below unless the
              module is the lowest one whose gradient is wanted (``_fused_stop_backward``)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, num_heads, scale, module):
        out, attn, zqk, xsc = ops.attention_forward_qkv(q, k, v, num_heads, scale, mask=mask, want_z=True,
                                                        want_x=mask is not None)
        ctx.save_for_backward(q, k, v, attn, out)      # (out: the softmax backward's row sums are d_out . out)
        ctx.num_heads, ctx.scale, ctx.module = num_heads, scale, module
        if xsc is None:
            xsc = zqk.new_empty(0)
        ctx.mark_non_differentiable(attn, zqk, xsc)
        ctx.set_materialize_grads(False)      # no [B,H,N,N] zero gradients for the by-products (vit._FusedAttention)
        return out, attn, zqk, xsc

    @staticmethod
    def backward(ctx, d_out, _a, _z, _x):
        q, k, v, attn, out = ctx.saved_tensors
        if d_out is None:
            return None, None, None, None, None, None, None
        stop = bool(getattr(ctx.module, "_fused_stop_backward", False))
        d_v = torch.empty_like(v)
        d_q = None if stop else torch.empty_like(q)
        d_k = None if stop else torch.empty_like(k)
        d_attn = ops.attention_backward_qkv(d_out, q, k, v, attn, ctx.num_heads, ctx.scale, d_q, d_k, d_v, need_qk=not stop, out=out)
        ctx.module.save_attn_gradients(d_attn)
        if stop:
            return None, None, None, None, None, None, None
        return d_q, d_k, d_v, None, None, None, None


def make_bert_module(L):
    ACT = {"relu": L.ReLU, "tanh": L.Tanh, "gelu": L.GELU}

    class BertEmbeddings(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size,
                                                padding_idx=getattr(config, "pad_token_id", 0))
            self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
            self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
            self.LayerNorm = L.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
            self.dropout = L.Dropout(config.hidden_dropout_prob)
            self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
            self.add1 = L.Add()
            self.add2 = L.Add()

        def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None):
            shape = input_ids.size() if input_ids is not None else inputs_embeds.size()[:-1]
            if position_ids is None:
                position_ids = self.position_ids[:, :shape[1]]
            if token_type_ids is None:
                token_type_ids = torch.zeros(shape, dtype=torch.long, device=self.position_ids.device)
            if inputs_embeds is None:
                inputs_embeds = self.word_embeddings(input_ids)
            pos = self.position_embeddings(position_ids).expand(shape[0], -1, -1)
            emb = self.add1([self.token_type_embeddings(token_type_ids), pos])
            emb = self.add2([emb, inputs_embeds])
            return self.dropout(self.LayerNorm(emb))

        def relprop(self, cam, **kwargs):        # BERT.py:87-94 (never called by BertModel.relprop)
            return self.add2.relprop(cam, **kwargs)

    class BertSelfAttention(nn.Module):
        def __init__(self, config):
            super().__init__()
            if config.hidden_size % config.num_attention_heads != 0:
                raise ValueError("hidden size must be a multiple of the number of attention heads")
            self.num_attention_heads = config.num_attention_heads
            self.attention_head_size = config.hidden_size // config.num_attention_heads
            self.all_head_size = config.hidden_size
            self.query = L.Linear(config.hidden_size, self.all_head_size)
            self.key = L.Linear(config.hidden_size, self.all_head_size)
            self.value = L.Linear(config.hidden_size, self.all_head_size)
            self.dropout = L.Dropout(config.attention_probs_dropout_prob)
            self.matmul1 = L.MatMul()
            self.matmul2 = L.MatMul()
            self.softmax = L.Softmax(dim=-1)
            self.add = L.Add()
            self.mul = L.Mul()
            self.clone = L.Clone()
            self.head_mask = self.attention_mask = None
            self.attn_cam = self.attn = self.attn_gradients = None

        def get_attn(self): return self.attn
        def save_attn(self, attn): self.attn = attn
        def save_attn_cam(self, cam): self.attn_cam = cam
        def get_attn_cam(self): return self.attn_cam
        def save_attn_gradients(self, g): self.attn_gradients = g
        def get_attn_gradients(self): return self.attn_gradients

        def transpose_for_scores(self, x):
            return x.view(*x.shape[:-1], self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

        def transpose_for_scores_relprop(self, x):
            return x.permute(0, 2, 1, 3).flatten(2)

        def forward(self, hidden_states, attention_mask=None, head_mask=None, **unused):
            if head_mask is not None:
                raise NotImplementedError("head_mask is off the accelerated path")
            self.head_mask, self.attention_mask = head_mask, attention_mask
            h1, h2, h3 = self.clone(hidden_states, 3)
            self._fused_anchor = None
            B, N, _ = hidden_states.shape
            if (ops.USE_FUSED_PRODUCERS and hidden_states.is_cuda and hidden_states.dtype == torch.float32
                    and not self.training and ops.attention_forward_supported(N, self.attention_head_size)
                    and (attention_mask is None or tuple(attention_mask.shape) == (B, 1, 1, N))):
                return self._forward_fused(h1, h2, h3, attention_mask)
            q = self.transpose_for_scores(self.query(h1))
            k = self.transpose_for_scores(self.key(h2))
            v = self.transpose_for_scores(self.value(h3))
            scores = self.matmul1([q, k.transpose(-1, -2)]) / math.sqrt(self.attention_head_size)
            if attention_mask is not None:
                scores = self.add([scores, attention_mask])
            probs = self.softmax(scores)
            self.save_attn(probs)
            if probs.requires_grad:
                probs.register_hook(self.save_attn_gradients)
            ctx = self.matmul2([self.dropout(probs), v])
            ctx = ctx.permute(0, 2, 1, 3).contiguous()
            return (ctx.view(*ctx.shape[:-2], self.all_head_size),)

        def _forward_fused(self, h1, h2, h3, attention_mask):
            """BERT.py:336-352 on the producer kernels.  The rule modules' caches (matmul1.X / .Y, add.X, matmul2.X / .Y)
            are views of the three Linear outputs and of the kernels' by-products, exactly the tensors the stock
            forward would have cached there (add.X[0] = the scaled scores WITHOUT the mask, BERT.py:339-342)."""
            B, N, C = h1.shape
            H, D = self.num_attention_heads, self.attention_head_size
            ql, kl, vl = self.query(h1), self.key(h2), self.value(h3)
            ctx, probs, zqk, xsc = _FusedSelfAttention.apply(ql, kl, vl, attention_mask, H, 1.0 / math.sqrt(D), self)
            self._fused_anchor = ql if ql.requires_grad else None
            heads = lambda t: t.detach().view(B, N, H, D).permute(0, 2, 1, 3)      # noqa: E731
            q, k, v = heads(ql), heads(kl), heads(vl)
            self.save_attn(probs)
            caches = [(self.matmul1, [q, k.transpose(-1, -2)], zqk), (self.matmul2, [probs, v], heads(ctx))]
            if attention_mask is not None:
                caches.append((self.add, [xsc, attention_mask], None))
            for mod, X, Y in caches:
                mod.X, mod.Y = X, Y
                mod._y_version = Y._version if Y is not None else None
                mod._w_version = (None, None)
            return (ctx,)

        def relprop(self, cam, **kwargs):
            """BERT.py:367-409.  cam [B,N,C] -> relevance of hidden_states [B,N,C]."""
            B, N, C = cam.shape
            H, D = self.num_attention_heads, self.attention_head_size
            var = self.matmul2.variant
            probs, v = self.matmul2.X
            q, kt = self.matmul1.X
            rq = torch.empty((B, N, C), dtype=cam.dtype, device=cam.device)
            rk, rv = torch.empty_like(rq), torch.empty_like(rq)
            as_heads = lambda t: t.view(B, N, H, D).permute(0, 2, 1, 3)          # noqa: E731  (views)
            cam1, _ = ops.matmul_relprop_av(as_heads(cam), probs, v, out_scale=0.5, cam_v_out=as_heads(rv), variant=var,
                                            z=R_ours._cached_y(self.matmul2))
            self.save_attn_cam(cam1)
            if getattr(self, "_stop_after_attn_cam", False):   # Generator(prune=True): nothing below is read
                raise L.StopRelprop()
            if self.attention_mask is not None:
                # (deferred: the Add's per-sample rescale rides with cam1 into the QK rule's S tile)
                cam1, _ = self.add.relprop(cam1, deferred=True, **kwargs)           # BERT.py:386-388
            ops.matmul_relprop_qk(cam1, q, kt.transpose(-1, -2), out_scale=0.5, cam_q_out=as_heads(rq),
                                  cam_k_out=as_heads(rk), variant=var, z=R_ours._cached_y(self.matmul1))
            rq = self.query.relprop(rq, **kwargs)
            rk = self.key.relprop(rk, **kwargs)
            rv = self.value.relprop(rv, **kwargs)
            return self.clone.relprop((rq, rk, rv), **kwargs)

    class BertSelfOutput(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.dense = L.Linear(config.hidden_size, config.hidden_size)
            self.LayerNorm = L.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
            self.dropout = L.Dropout(config.hidden_dropout_prob)
            self.add = L.Add()

        def forward(self, hidden_states, input_tensor):
            return self.LayerNorm(self.add([self.dropout(self.dense(hidden_states)), input_tensor]))

        def relprop(self, cam, **kwargs):        # BERT.py:427-434
            # (deferred: the Add's per-sample rescale rides with cam1 / cam2 into dense.relprop / the Clone rule)
            cam1, cam2 = self.add.relprop(cam, deferred=True, **kwargs)
            return self.dense.relprop(cam1, **kwargs), cam2

    class BertAttention(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.self = BertSelfAttention(config)
            self.output = BertSelfOutput(config)
            self.clone = L.Clone()

        def forward(self, hidden_states, attention_mask=None, head_mask=None, **unused):
            h1, h2 = self.clone(hidden_states, 2)
            return (self.output(self.self(h1, attention_mask, head_mask)[0], h2),)

        def relprop(self, cam, **kwargs):        # BERT.py:240-247
            cam1, cam2 = self.output.relprop(cam, **kwargs)
            cam1 = self.self.relprop(cam1, **kwargs)
            return self.clone.relprop((cam1, cam2), **kwargs)

    class BertIntermediate(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.dense = L.Linear(config.hidden_size, config.intermediate_size)
            act = config.hidden_act
            self.intermediate_act_fn = ACT[act]() if isinstance(act, str) else act

        def forward(self, hidden_states):
            return self.intermediate_act_fn(self.dense(hidden_states))

        def relprop(self, cam, **kwargs):        # BERT.py:451-456 (activation rule = identity)
            return self.dense.relprop(cam, **kwargs)

    class BertOutput(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.dense = L.Linear(config.intermediate_size, config.hidden_size)
            self.LayerNorm = L.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
            self.dropout = L.Dropout(config.hidden_dropout_prob)
            self.add = L.Add()

        def forward(self, hidden_states, input_tensor):
            return self.LayerNorm(self.add([self.dropout(self.dense(hidden_states)), input_tensor]))

        def relprop(self, cam, **kwargs):        # BERT.py:474-487
            cam1, cam2 = self.add.relprop(cam, deferred=True, **kwargs)
            return self.dense.relprop(cam1, **kwargs), cam2

    class BertLayer(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.attention = BertAttention(config)
            self.intermediate = BertIntermediate(config)
            self.output = BertOutput(config)
            self.clone = L.Clone()
            act = self.intermediate.intermediate_act_fn
            if hasattr(act, "feeds"):
                act.feeds(self.output.dense)      # (a hint for the producers: GELU may emit output.dense's operand planes)

        def forward(self, hidden_states, attention_mask=None, head_mask=None, **unused):
            att = self.attention(hidden_states, attention_mask, head_mask)[0]
            a1, a2 = self.clone(att, 2)
            return (self.output(self.intermediate(a1), a2),)

        def relprop(self, cam, **kwargs):        # BERT.py:521-530
            cam1, cam2 = self.output.relprop(cam, **kwargs)
            cam1 = self.intermediate.relprop(cam1, **kwargs)
            cam = self.clone.relprop((cam1, cam2), **kwargs)
            return self.attention.relprop(cam, **kwargs)

        def relprop_cls_only(self, cam_cls, **kwargs):
            """BertLayer.relprop for relevance that lives on token 0 only (cam_cls [B,1,C]) -- the state right after
            BertPooler.relprop (BERT.py:181-191), i.e. the LAST layer.  Every rule from output.add down to
            attention.output.dense maps a zero relevance row to an exact zero row and Add's per-sample sums gain only
            zeros from those rows, so these rules run on the [B,1,C] slice of their cached inputs; the result is
            scattered into zero [B,N,C] tensors before the self-attention rules, which spread relevance to all
            tokens.  Same values as the dense evaluation at 1/N of its Linear work (see vit.Block.relprop_cls_only)."""
            alpha = kwargs.get("alpha", 1)
            var = self.clone.variant
            cls = lambda t: t[:, :1]                                             # noqa: E731
            def lin(r, m):       # the staleness guard of the cached forward output applies here as in Linear.relprop
                y = R_ours._cached_y(m)
                return ops.linear_relprop(r, cls(m.X), m.weight.detach(), alpha=alpha, variant=var,
                                          Y=None if y is None else cls(y), bias=m.bias, cache=R_ours.x6_cache(m))
            dfr = ops.USE_DEFERRED_ADD
            c1, c2 = ops.add_relprop(cam_cls, cls(self.output.add.X[0]), cls(self.output.add.X[1]), variant=var,
                                     deferred=dfr)
            c1 = lin(lin(c1, self.output.dense), self.intermediate.dense)
            cam = ops.clone_relprop((c1, c2), cls(self.clone.X))
            att = self.attention
            a1, a2 = ops.add_relprop(cam, cls(att.output.add.X[0]), cls(att.output.add.X[1]), variant=var, deferred=dfr)
            a1 = lin(a1, att.output.dense)
            B, N, C = att.clone.X.shape
            dense = torch.zeros((2, B, N, C), dtype=a1.dtype, device=a1.device)
            dense[0, :, 0] = a1[:, 0]
            if isinstance(a2, ops.Deferred):
                a2 = a2.materialise()
            dense[1, :, 0] = a2[:, 0]
            cam1 = att.self.relprop(dense[0], **kwargs)
            return att.clone.relprop((cam1, dense[1]), **kwargs)

    class BertEncoder(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

        def forward(self, hidden_states, attention_mask=None, head_mask=None, **unused):
            for i, layer in enumerate(self.layer):
                hidden_states = layer(hidden_states, attention_mask, None if head_mask is None else head_mask[i])[0]
            return (hidden_states,)

        def relprop(self, cam, **kwargs):        # BERT.py:155-159
            for layer in reversed(self.layer):
                cam = layer.relprop(cam, **kwargs)
            return cam

    class BertPooler(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.dense = L.Linear(config.hidden_size, config.hidden_size)
            self.activation = L.Tanh()
            self.pool = L.IndexSelect()
            self.register_buffer("_cls_index", torch.zeros((), dtype=torch.long), persistent=False)

        def forward(self, hidden_states):
            first = self.pool(hidden_states, 1, self._cls_index).squeeze(1)
            return self.activation(self.dense(first))

        def relprop(self, cam, **kwargs):        # BERT.py:181-191 (tanh rule = identity)
            cam = self.dense.relprop(cam, **kwargs)
            return self.pool.relprop(cam.unsqueeze(1), **kwargs)

    class BertModel(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embeddings = BertEmbeddings(config)
            self.encoder = BertEncoder(config)
            self.pooler = BertPooler(config)

        def get_input_embeddings(self):
            return self.embeddings.word_embeddings

        def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                    head_mask=None, inputs_embeds=None, **unused):
            shape = input_ids.size() if input_ids is not None else inputs_embeds.size()[:-1]
            device = input_ids.device if input_ids is not None else inputs_embeds.device
            if attention_mask is None:
                attention_mask = torch.ones(shape, device=device)
            # transformers 3.5.1 get_extended_attention_mask: (1 - mask)[:, None, None, :] * -10000
            ext = (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0
            emb = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids,
                                  inputs_embeds=inputs_embeds)
            seq = self.encoder(emb, attention_mask=ext, head_mask=head_mask)[0]
            return (seq, self.pooler(seq))

        exploit_cls_sparsity = True   # exact; set False to evaluate the last layer densely

        def relprop(self, cam, **kwargs):        # BERT.py:645-651
            if not self.exploit_cls_sparsity:
                return self.encoder.relprop(self.pooler.relprop(cam, **kwargs), **kwargs)
            # pooler.relprop puts relevance on token 0 only: keep it as a [B,1,C] row through the last layer's
            # dense rules instead of a [B,N,C] tensor that is zero everywhere else
            cam = self.pooler.dense.relprop(cam, **kwargs)
            cam = ops.index_select_relprop(cam.unsqueeze(1), self.pooler.pool.X[:, :1], 0)
            layers = list(self.encoder.layer)
            cam = layers[-1].relprop_cls_only(cam, **kwargs)
            for layer in reversed(layers[:-1]):
                cam = layer.relprop(cam, **kwargs)
            return cam

    class BertForSequenceClassification(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.num_labels = config.num_labels
            self.bert = BertModel(config)
            self.dropout = L.Dropout(config.hidden_dropout_prob)
            self.classifier = L.Linear(config.hidden_size, config.num_labels)
            self.apply(self._init_weights)

        def _init_weights(self, m):
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=getattr(self.config, "initializer_range", 0.02))
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

        def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                    head_mask=None, inputs_embeds=None, labels=None, **unused):
            pooled = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                               position_ids=position_ids, head_mask=head_mask, inputs_embeds=inputs_embeds)[1]
            logits = self.classifier(self.dropout(pooled))
            if labels is not None:
                if self.num_labels == 1:
                    loss = nn.functional.mse_loss(logits.view(-1), labels.view(-1))
                else:
                    loss = nn.functional.cross_entropy(logits.view(-1, self.num_labels), labels.view(-1))
                return (loss, logits)
            return (logits,)

        def relprop(self, cam=None, **kwargs):   # BertForSequenceClassification.py:83-88
            cam = self.classifier.relprop(cam, **kwargs)
            return self.bert.relprop(cam, **kwargs)

    return dict(BertEmbeddings=BertEmbeddings, BertSelfAttention=BertSelfAttention, BertSelfOutput=BertSelfOutput,
                BertAttention=BertAttention, BertIntermediate=BertIntermediate, BertOutput=BertOutput,
                BertLayer=BertLayer, BertEncoder=BertEncoder, BertPooler=BertPooler, BertModel=BertModel,
                BertForSequenceClassification=BertForSequenceClassification)


_ns = make_bert_module(R_ours)
BertModel = _ns["BertModel"]
BertForSequenceClassification = _ns["BertForSequenceClassification"]
