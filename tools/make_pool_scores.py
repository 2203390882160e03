"""Regenerate tests/golden/pool_scores.npz: the float64 per-frame scores bench.py's `score_digest` is compared with.

GPU box:  python tools/make_pool_scores.py   ->  gpurun_out/pool_scores.npz   (copy it to tests/golden/ afterwards)

Keys (bench.table_key): <model>_c<channels>k<classes>_<H>x<W>_<measure>_seed<weights seed>
  enet_c3k19_1024x2048_entropy_seed0   all 2975 frames of the synthetic pool   (BASELINE configs[1] / [2])
  icnet_c3k19_1024x2048_margin_seed0   all 2975 frames                        (configs[3])
  enet_c4k6_1024x2048_entropy_seed1    all 2975 frames                        (configs[4])
The scores are per-image float64 means produced by the HIP path (bitwise reproducible, independent of the batch
composition: tests/test_gpu_parity.py::test_score_is_bitwise_reproducible).  What ties the table to the oracle: the
full-resolution parity tests (test_full_resolution_image_bit_exact, test_full_resolution_c5_rgb_nir_frame_bit_exact,
test_icnet_gpu.py::test_full_resolution_frame_bit_exact) check frames of it against the C oracle (logits / labels
bit-exact, mean <= 1e-6) AND assert that the HIP score of that frame equals the table entry bit for bit;
test_pool_score_table_* re-scores further entries in other batch compositions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import semanticsegmentationactivelearning_amd as ssal
from semanticsegmentationactivelearning_amd import synthetic as syn

H, W, BS = 1024, 2048, 8
out = {}
for model, c, k, measure, seed, count in (("enet", 3, 19, "entropy", 0, bench.POOL), ("icnet", 3, 19, "margin", 0, bench.POOL),
                                          ("enet", 4, 6, "entropy", 1, bench.POOL)):
    net = ssal.ICNet(k) if model == "icnet" else ssal.ENet(k)
    net.build((None, None, None, c))
    (syn.randomize_icnet if model == "icnet" else syn.randomize_enet)(net, seed=seed)
    scores = []
    for first in range(0, count, BS):
        n = min(BS, count - first)
        x = syn.synth_frames_device(first, n, H, W, c)
        scores.append(net.score(x, measure=measure).cpu().numpy())
    key = bench.table_key(model, c, k, H, W, measure, seed)
    out[key] = np.concatenate(scores).astype(np.float64)
    print(key, out[key].shape, out[key][:3], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "pool_scores.npz"), **out)
print("wrote gpurun_out/pool_scores.npz")
