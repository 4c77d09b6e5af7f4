"""How long after a synchronised start does the GPU begin a replayed step?

    python scripts/graph_fill_probe.py          # on a GPU box

Times, for the cfg-2 train step replayed from its HIP graph: the host duration of one `graph.replay()` call, the wall time of ONE synchronised step, and of
trains of 10 and 40 - the difference between one synchronised step and the steady step is what a caller who synchronises every step pays per step.
Round 5 (gpurun_out/r05_call24 had shown bench.py's 10-step runs 1.5-2 % slower per step than its 80-step runs, whatever the warm-up): the replay call
takes 0.10 ms of host time and a synchronised single step costs 0.07 ms more than a step inside a train - the graph launch is NOT what short runs pay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from biapy_amd.graphs import GraphedTrainStep
    from biapy_amd.losses import BCEWithLogitsLoss
    from biapy_amd.resunet import ResUNet
    torch.manual_seed(0)
    model = ResUNet(image_shape=(128, 128, 128, 1), activation="elu", feature_maps=bench.FM, drop_values=[0.0] * 5, normalization="in",
                    yx_down=[2] * 4, z_down=[2] * 4, isotropy=[True] * 5, larger_io=False, conv_layers=[2] * 5, compute_dtype=torch.float16).to(dev)
    model.train()
    x, t = bench.synth_batch(4, 128, dev, seed=0)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    step = GraphedTrainStep(model, BCEWithLogitsLoss(), opt, x, t)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for label, fn in (("train step", step),):
        one = []
        host = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e3)
            one.append((t2 - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ten = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        forty = (time.perf_counter() - t0) * 1e3
        print(f"{label}: host time of one call {min(host):.3f} ms (median {sorted(host)[4]:.3f}); one synchronised step {min(one):.3f} ms (median {sorted(one)[4]:.3f}); "
              f"10 steps {ten:.3f} ms = {ten / 10:.3f} each; 40 steps {forty / 40:.3f} each; steady step from the two trains {(forty - ten) / 30:.3f} ms; "
              f"fixed cost of a synchronised start {ten - 10 * (forty - ten) / 30:.3f} ms")


if __name__ == "__main__":
    main()
