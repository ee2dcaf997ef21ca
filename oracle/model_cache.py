"""Adapters that collect the tensors relprop reads from OUR models (after forward + attention
gradients) into the dict layout the CPU oracle consumes (oracle/relprop_oracle.py).

TEST INFRASTRUCTURE ONLY (see oracle/relprop_oracle.py header): used by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg to run the oracle on exactly the activations the HIP path consumed.
"""


def _cpu(t):
    return None if t is None else t.detach().float().cpu()


def vit_cache_from_model(model):
    """Cached forward tensors of OUR ViT (after fwd + attention gradients), oracle naming."""
    blocks = []
    for blk in model.blocks:
        blocks.append({
            "add2_x0": _cpu(blk.add2.X[0]), "add2_x1": _cpu(blk.add2.X[1]),
            "fc2_x": _cpu(blk.mlp.fc2.X), "fc2_w": _cpu(blk.mlp.fc2.weight),
            "fc1_x": _cpu(blk.mlp.fc1.X), "fc1_w": _cpu(blk.mlp.fc1.weight),
            "clone2_x": _cpu(blk.clone2.X), "add1_x0": _cpu(blk.add1.X[0]), "add1_x1": _cpu(blk.add1.X[1]),
            "proj_x": _cpu(blk.attn.proj.X), "proj_w": _cpu(blk.attn.proj.weight),
            "attn": _cpu(blk.attn.get_attn()), "attn_grad": _cpu(blk.attn.get_attn_gradients()),
            "qkv_out": _cpu(blk.attn.qkv.Y), "qkv_x": _cpu(blk.attn.qkv.X), "qkv_w": _cpu(blk.attn.qkv.weight),
            "clone1_x": _cpu(blk.clone1.X),
            # the attention products as the forward pass computed them (what the reference's autograd re-evaluation
            # reproduces on that device): unscaled q k^T and attn v
            "z_qk": _cpu(getattr(blk.attn.matmul1, "Y", None)), "z_av": _cpu(getattr(blk.attn.matmul2, "Y", None))})
    return {"head_x": _cpu(model.head.X), "head_w": _cpu(model.head.weight), "pool_x": _cpu(model.pool.X),
            "blocks": blocks,
            # method="full" only: position-embedding Add and the patch-embedding convolution
            "pos_add_x0": _cpu(model.add.X[0]), "pos_embed": _cpu(model.add.X[1]),
            "patch_x": _cpu(model.patch_embed.proj.X), "patch_w": _cpu(model.patch_embed.proj.weight)}


def bert_cache_from_model(model):
    import math
    layers = []
    for lay in model.bert.encoder.layer:
        sa = lay.attention.self
        masked = sa.attention_mask is not None
        # The first operand of the mask Add (BERT.py:339-342: the scaled scores BEFORE the mask is added) is RECOMPUTED here
        # from the cached q k^T instead of read from the module's own add.X[0]: a producer that caches the wrong tensor
        # there (scores + mask, i.e. the mask twice in Add.relprop's denominator) then disagrees with the oracle.
        zqk = _cpu(getattr(sa.matmul1, "Y", None))
        x0 = None
        if masked:
            x0 = zqk / math.sqrt(sa.attention_head_size) if zqk is not None else _cpu(sa.add.X[0])
        layers.append({
            "out_add_x0": _cpu(lay.output.add.X[0]), "out_add_x1": _cpu(lay.output.add.X[1]),
            "out_dense_x": _cpu(lay.output.dense.X), "out_dense_w": _cpu(lay.output.dense.weight),
            "inter_x": _cpu(lay.intermediate.dense.X), "inter_w": _cpu(lay.intermediate.dense.weight),
            "clone_x": _cpu(lay.clone.X),
            "att_add_x0": _cpu(lay.attention.output.add.X[0]), "att_add_x1": _cpu(lay.attention.output.add.X[1]),
            "att_dense_x": _cpu(lay.attention.output.dense.X), "att_dense_w": _cpu(lay.attention.output.dense.weight),
            "probs": _cpu(sa.get_attn()), "attn_grad": _cpu(sa.get_attn_gradients()),
            "q": _cpu(sa.query.Y), "k": _cpu(sa.key.Y), "v": _cpu(sa.value.Y),
            "mask_add_x0": x0,
            "ext_mask": _cpu(sa.add.X[1]) if masked else None,
            "q_x": _cpu(sa.query.X), "q_w": _cpu(sa.query.weight), "k_x": _cpu(sa.key.X), "k_w": _cpu(sa.key.weight),
            "v_x": _cpu(sa.value.X), "v_w": _cpu(sa.value.weight),
            "self_clone_x": _cpu(sa.clone.X), "att_clone_x": _cpu(lay.attention.clone.X),
            "z_qk": zqk, "z_av": _cpu(getattr(sa.matmul2, "Y", None))})
    return {"cls_x": _cpu(model.classifier.X), "cls_w": _cpu(model.classifier.weight),
            "pool_dense_x": _cpu(model.bert.pooler.dense.X), "pool_dense_w": _cpu(model.bert.pooler.dense.weight),
            "pool_x": _cpu(model.bert.pooler.pool.X), "layers": layers}


class sliced_relprop_state:
    """Context manager: temporarily replace every cached tensor with leading dimension `B` (module attributes
    X / Y / attn / gradients / masks) by its slice [i:i+1], so that relprop runs on sample i alone -- on exactly
    the tensors the batched run consumed."""

    def __init__(self, model, i, B):
        self.model, self.i, self.B = model, i, B
        self.saved = []

    def _slice(self, v):
        import torch
        if torch.is_tensor(v) and v.dim() >= 2 and v.shape[0] == self.B:
            return v[self.i:self.i + 1]
        if isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) for t in v):
            return type(v)(self._slice(t) for t in v)
        return v

    def __enter__(self):
        for m in self.model.modules():
            for name, val in list(vars(m).items()):
                if name.startswith("_"):
                    continue
                new = self._slice(val)
                if new is not val:
                    self.saved.append((m, name, val))
                    setattr(m, name, new)
        return self

    def __exit__(self, *a):
        for m, name, val in self.saved:
            setattr(m, name, val)
