"""End-to-end rate of the ranking pass fed from REAL TFRecords (one PNG-encoded 1024x2048x3 example per file, the
reference's schema): how fast tensortools.InputStage decodes with N workers, and what the GPU path delivers behind it
(uint8 frames in page-locked batches, side-stream copy, on-GPU conversion + scoring).  The bench line of bench.py keeps
the frames resident in HBM (as the task's measurement contract asks); this script measures the system around it.
Usage: python tools/input_bench.py [frames=64] [height=1024] [width=2048]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from semanticsegmentationactivelearning_amd import synthetic as syn
from semanticsegmentationactivelearning_amd.tensortools import InputStage, tfrecord


def png(arr):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr[..., 0] if arr.shape[-1] == 1 else arr).save(b, format="PNG")
    return b.getvalue()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    tmp = tempfile.mkdtemp(prefix="ssal_pool_")
    t0 = time.perf_counter()
    nbytes = 0
    for i in range(n):
        img = syn.synth_frame_u8(i, h, w, 3)
        data = png(img)
        nbytes += len(data)
        feats = {"image/data": data, "image/encoding": "png", "image/channels": 3, "label": b"",
                 "height": h, "width": w, "id": "frame_%04d" % i}
        tfrecord.write_tfrecord(os.path.join(tmp, "frame_%04d.tfrecord" % i), [tfrecord.make_example(feats)])
    print("wrote %d records %dx%d, %.1f MB PNG each, in %.1f s (cores usable: %d)"
          % (n, h, w, nbytes / n / 1e6, time.perf_counter() - t0, len(os.sched_getaffinity(0))), flush=True)

    for workers in (1, 4, 8, 15):
        stage = InputStage(input_shape=[h, w], workers=workers, image_dtype=np.uint8)
        stage.add_dataset("val", tmp, batch_size=8)
        stage.init_iterator("val")
        t0 = time.perf_counter()
        cnt = 0
        for image, label, mask in stage:
            cnt += len(image)
        dt = time.perf_counter() - t0
        print("decode only, %2d workers: %6.1f images/s" % (workers, cnt / dt), flush=True)

    import torch
    if not torch.cuda.is_available():
        return
    import semanticsegmentationactivelearning_amd as ssal
    from semanticsegmentationactivelearning_amd import active_learning as al
    net = ssal.ENet(19)
    net.build((None, None, None, 3))
    syn.randomize_enet(net, seed=0)
    for workers in (8, 15):
        stage = InputStage(input_shape=[h, w], workers=workers, image_dtype=np.uint8, pin_memory=True, pin_buffers=6)
        stage.add_dataset("val", tmp, batch_size=8)
        for rep in range(2):  # first pass warms the page-locked ring and the kernels
            stage.init_iterator("val")
            pos = [0]

            def batches():
                for image, label, mask in stage:
                    k = len(image)
                    yield image, np.arange(pos[0], pos[0] + k)
                    pos[0] += k

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            low, uc = al.rank_confidence(net, batches(), n, np.arange(n), min(8, n), measure="entropy", prefetch=2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("TFRecords -> decode (%2d workers) -> pinned uint8 -> GPU score -> rank: %6.1f images/s" % (workers, n / dt),
              flush=True)


if __name__ == "__main__":
    main()
