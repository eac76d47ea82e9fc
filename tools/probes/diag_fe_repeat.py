"""Run the backbone's forward + backward plan several times on fresh FeatureExtractor instances with identical inputs and report which
activations / parameter gradients differ between runs (kernel determinism at the small test geometry, 128x192, batch 2)."""
import importlib
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
FE = importlib.import_module("2d_object_detection_amd.models.feature_extractor")
RT = importlib.import_module("2d_object_detection_amd.runtime")
BF = torch.bfloat16


def run_once(shape=(128, 192, 3), batch=2):
    fe = FE.FeatureExtractor(shape, depth=50, device="cuda")
    g = torch.Generator().manual_seed(5)
    for u in fe.conv_units():
        fe.store.weight(u.name + "_bn/gamma").copy_((torch.rand(u.cout, generator=g) + 0.5) * (0.25 if u.name.endswith("_3") else 1.0))
    fe.setup(batch, True)
    fe.images.copy_(torch.randint(0, 256, (batch,) + shape, generator=g, dtype=torch.uint8))
    fe.store.refresh_bf16()
    _, gh, gw, cf = fe.output_shape
    g_feat = (torch.randn(batch * gh * gw, cf, generator=g) * 1e-2).to(BF).cuda()
    plan = RT.Plan("backbone")
    plan.zero(fe.store.g)
    fe.refresh_weights(plan)
    fe.forward_plan(plan, True)
    fe.backward_plan(plan, g_feat, g_feat_reduced=False)
    junk = torch.randn(32 * 1024 * 1024, device="cuda") * float("nan")      # poison freed memory for the next instance's torch.empty
    del junk
    plan.run()
    torch.cuda.synchronize()
    out = {"feat": fe.feature_maps.float().clone()}
    for u in fe.conv_units():
        out["z:" + u.name] = u.z.float().clone()
        out["dz:" + u.name] = u.dz.float().clone()
        out["gw:" + u.name] = fe.store.grad(u.name + "_conv/kernel").clone()
    return out


def main():
    runs = [run_once() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
    ref = runs[0]
    bad = 0
    for i, r in enumerate(runs[1:], 1):
        for k in ref:
            a, b = ref[k], r[k]
            if not torch.equal(a, b):
                rel = float((a - b).norm() / (a.norm() + 1e-30))
                nan = int(torch.isnan(b).sum())
                if rel > 1e-6 or nan:
                    bad += 1
                    print("run %d %-32s rel %.3e nan %d" % (i, k, rel, nan), flush=True)
    print("differences above 1e-6:", bad)


if __name__ == "__main__":
    main()
