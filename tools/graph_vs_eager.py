"""Bisect aid: the config-2 clip run eagerly twice and replayed three times; prints which outputs (and, with TCE_TAPS=1,
which intermediates of the encoder) differ -- a race between graph branches shows as eager != replay or replay != replay."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tce_rvos_amd import build_model, load_synth_weights
from tce_rvos_amd._lib import lib
if os.environ.get("FEWQ") == "0":
    lib().tce_debug_msda_set_fewq(0)
ns = argparse.Namespace(backbone="swin_t_p4w7", with_box_refine=True, binary=True, freeze_text_encoder=True, f_token=8, qtrans=True,
                        num_feature_levels=4, text_encoder_layers=1)
model, _, _ = build_model(ns)
model = model.cuda().eval()
load_synth_weights(model, 11)
model.repack()
g = torch.Generator().manual_seed(1)
T, H, W = 5, 360, 640
frames = torch.randn(T, 3, H, W, generator=g).cuda()
hid = torch.randn(32, 768, generator=g).cuda()
pooled = torch.tanh(torch.randn(768, generator=g)).cuda()
model.use_graph = False
e = [model.forward_features(frames, hid, pooled, float(H), float(W)) for _ in range(2)]
model.use_graph = True
r = [model.forward_features(frames, hid, pooled, float(H), float(W)) for _ in range(6)][1:]
torch.cuda.synchronize()
for k in ("memory", "pred_logits", "pred_boxes", "pred_masks"):
    d = [(x[k] - e[0][k]).abs().max().item() for x in [e[1]] + r]
    print(f"{k:12s} eager2-eager1 {d[0]:.3e}   replays - eager1: " + " ".join(f"{v:.3e}" for v in d[1:]))
if "taps" in e[0]:
    for k in e[0]["taps"][0]:
        d = [(x["taps"][0][k] - e[0]["taps"][0][k]).abs().max().item() for x in r]
        print(f"  tap {k:12s} replays - eager1: " + " ".join(f"{v:.3e}" for v in d))
