"""segmentation metrics of a relevance map vs the reference's utils/metrices.py functions called the
way imagenet_seg_eval.py:229-268 calls them (tests/golden/seg_metrics.npz, make_golden.make_segmentation)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def inputs():
    g = torch.Generator().manual_seed(9)                      # = make_golden.segmentation_inputs
    heat = (torch.rand((3, 32, 32), generator=g) * 50).round() / 50
    heat[0, :4] = 0.5
    mask = (heat > heat.flatten(1).mean(1).view(-1, 1, 1)).float()
    labels = (torch.rand((3, 32, 32), generator=g) > 0.6).long()
    labels[2, 5] = 0
    mask[2, 5] = 0
    return heat, mask, labels


def _check(device):
    from transformer_explainability_amd import segmentation as sg
    g = load_golden("seg_metrics.npz")
    heat, mask, labels = (t.to(device) for t in inputs())
    ev = sg.SegmentationEvaluator(explain=None)
    correct, labeled, inter, union, ap, f1 = ev.update_from_heat(heat, mask, labels)
    assert torch.equal(correct.cpu(), g["correct"].long()) and torch.equal(labeled.cpu(), g["labeled"].long())
    assert torch.equal(inter.cpu(), g["inter"].long()) and torch.equal(union.cpu(), g["union"].long())
    assert float((ap.cpu() - g["ap"]).abs().max()) < 1e-12
    assert float((f1.cpu() - g["f1"]).abs().max()) < 1e-12
    s = ev.summary()
    eps = np.spacing(1, dtype=np.float64)
    assert abs(s["pixAcc"] - g["correct"].sum().item() / (eps + g["labeled"].sum().item())) < 1e-12
    iou = g["inter"].sum(0).double() / (eps + g["union"].sum(0).double())
    assert abs(s["mIoU"] - float(iou.mean())) < 1e-12
    assert abs(s["mAP"] - float(g["ap"].mean())) < 1e-12 and abs(s["mF1"] - float(g["f1"].mean())) < 1e-12
    # ignore labels drop out of accuracy / IoU / AP
    lab2 = labels.clone()
    lab2[:, :3] = -1
    c2, l2 = sg.pixel_accuracy(mask, lab2)
    assert int(l2[0]) == 32 * 29 and (c2 <= l2).all()
    assert torch.isfinite(sg.average_precision(heat, lab2)).all()


def test_metrics_match_reference_cpu():
    _check(torch.device("cpu"))


def test_average_precision_is_sklearns():
    sk = pytest.importorskip("sklearn.metrics")
    from transformer_explainability_amd import segmentation as sg
    g = torch.Generator().manual_seed(4)
    heat = (torch.rand((2, 16, 16), generator=g) * 10).round() / 10          # many ties
    labels = (torch.rand((2, 16, 16), generator=g) > 0.5).long()
    ap = sg.average_precision(heat, labels)
    for b in range(2):
        p = torch.cat([1 - heat[b].flatten(), heat[b].flatten()]).numpy()
        t = torch.cat([labels[b].flatten() == 0, labels[b].flatten() == 1]).numpy().astype(np.int64)
        assert abs(float(ap[b]) - sk.average_precision_score(t, p)) < 1e-12


def test_evaluator_end_to_end_cpu():
    """explain -> te_heatmap (oracle here) -> metrics on a tiny ViT: runs, shapes, ranges."""
    from oracle_backend import oracle_ops
    from transformer_explainability_amd import segmentation as sg, vit
    from transformer_explainability_amd.generators import LRP
    torch.manual_seed(0)
    m = vit.VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=4, num_classes=10,
                              qkv_bias=True).eval()
    x = torch.randn(3, 3, 32, 32)
    labels = (torch.rand(3, 32, 32) > 0.5).long()
    with oracle_ops():
        lrp = LRP(m)
        ev = sg.SegmentationEvaluator(lambda im: lrp.generate_LRP(im, start_layer=1), scale=8)
        ev.update(x, labels)
    s = ev.summary()
    assert all(0.0 <= v <= 1.0 for v in s.values()) and len(ev.total_ap) == 3 and len(ev.total_f1) == 3


@pytest.mark.gpu
def test_metrics_match_reference_gpu():
    _check(torch.device("cuda:0"))
