"""The reference's own import lines must resolve to this implementation after install_dropin():
bare module names used by the scripts that run from baselines/ViT, package paths used by the BERT pipeline."""
import subprocess
import sys

IMPORTS = r'''
import sys
sys.path.insert(0, %r)
import transformer_explainability_amd as te
te.install_dropin()
from ViT_explanation_generator import Baselines, LRP                      # imagenet_seg_eval.py:19
from ViT_new import vit_base_patch16_224                                  # imagenet_seg_eval.py:20
from ViT_LRP import vit_base_patch16_224 as vit_LRP                       # imagenet_seg_eval.py:21
from ViT_orig_LRP import vit_base_patch16_224 as vit_orig_LRP             # imagenet_seg_eval.py:22
from baselines.ViT.ViT_LRP import vit_base_patch16_224 as a, vit_large_patch16_224 as b, deit_base_patch16_224 as c
from baselines.ViT.ViT_explanation_generator import LRP as L2
from BERT_explainability.modules.BERT.ExplanationGenerator import Generator            # bert_pipeline.py:17
from BERT_explainability.modules.BERT.BertForSequenceClassification import BertForSequenceClassification
from BERT_explainability.modules.BERT.BERT import BertModel
from BERT_explainability.modules.layers_ours import Linear as BL, MatMul, Tanh
from dataset.expl_hdf5 import ImagenetResults                            # pertubation_eval_from_hdf5.py:13
import modules.layers_ours as lo, modules.layers_lrp as ll
assert lo.Linear.variant == "ours" and ll.Linear.variant == "lrp"
for name in ['forward_hook', 'Clone', 'Add', 'Cat', 'ReLU', 'GELU', 'Dropout', 'BatchNorm2d', 'Linear', 'MaxPool2d',
             'AdaptiveAvgPool2d', 'AvgPool2d', 'Conv2d', 'Sequential', 'safe_divide', 'einsum', 'Softmax',
             'IndexSelect', 'LayerNorm', 'AddEye']:                          # layers_ours.py:5-7 (__all__)
    assert hasattr(lo, name) and hasattr(ll, name), name
m = vit_LRP()
assert m.default_method == "transformer_attribution" and vit_orig_LRP().default_method == "grad"
import inspect
sig = inspect.signature(LRP.generate_LRP)
assert list(sig.parameters)[:6] == ["self", "input", "index", "method", "is_ablation", "start_layer"]
sig = inspect.signature(Generator.generate_LRP)
assert list(sig.parameters)[:5] == ["self", "input_ids", "attention_mask", "index", "start_layer"]
assert sig.parameters["start_layer"].default == 11
print("ok")
'''


def test_reference_import_lines_resolve():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", IMPORTS % root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_baselines_cam_attn_on_cpu():
    """Baselines.generate_cam_attn needs no device kernel: shapes and the [0,1] range (ViT_explanation_generator.py:50-72)."""
    import torch
    from transformer_explainability_amd import vit
    from transformer_explainability_amd.generators import Baselines
    torch.manual_seed(0)
    model = vit.VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=4, num_classes=10,
                                  qkv_bias=True).eval()
    x = torch.randn(3, 3, 32, 32)
    cam = Baselines(model).generate_cam_attn(x)
    assert cam.shape == (3, 4, 4)
    for c in cam:                                # a map clamped to all-zero is 0/0 there, as in the reference
        assert bool(torch.isnan(c).all()) or (float(c.min()) == 0.0 and float(c.max()) == 1.0)
    one = Baselines(model).generate_cam_attn(x[:1], index=2)
    assert one.shape == (4, 4)
