#!/usr/bin/env python3
"""Generate the committed golden vectors from the fp32 CPU oracle.

    python tests/golden/make_golden.py

Writes tests/golden/*.pt (a few hundred KB in total). Every file holds seeded INPUTS and the
oracle's fp32 OUTPUT so the GPU tests can check the HIP path without the oracle, and the CPU tests
can detect drift of the oracle itself. The reference's own tests keep no stored vectors (they are
differential), and the reference cannot be imported here (needs its CUDA
extension), so these are minted from the restatement -- see the "parity unpinned" note in
oracle/unet_ref.py.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ops_ref as R  # noqa: E402
from oracle import unet_ref as U  # noqa: E402
from oracle import vae_ref as V  # noqa: E402


def g(seed):
    return torch.Generator().manual_seed(seed)


def h(t):  # round to fp16 so the fixture inputs are exactly representable in the kernels' I/O dtype
    return t.half().float()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}

    # GroupNorm(+SiLU): the reference self-test shape is randn(2,320,32,32) G=32 (group_norm.py:485-523);
    # the fixture keeps G=32, C=320 (10 channels per group) at 4x4 to stay small.
    x = h(torch.randn(2, 320, 4, 4, generator=g(1)) * 2 + 0.5)
    w, b = h(1 + 0.1 * torch.randn(320, generator=g(2))), h(0.1 * torch.randn(320, generator=g(3)))
    out["group_norm_silu"] = dict(x=x, weight=w, bias=b, groups=32, eps=1e-5, y=R.group_norm_ref(x, 32, w, b, 1e-5, True))
    out["group_norm"] = dict(x=x, weight=w, bias=b, groups=32, eps=1e-6, y=R.group_norm_ref(x, 32, w, b, 1e-6, False))

    x = h(torch.randn(37, 320, generator=g(4)))
    w, b = h(1 + 0.1 * torch.randn(320, generator=g(5))), h(0.1 * torch.randn(320, generator=g(6)))
    out["layer_norm"] = dict(x=x, weight=w, bias=b, eps=1e-5, y=R.layer_norm_ref(x, (320,), w, b))

    x = h(torch.randn(48, 64, generator=g(7)))
    w, b = h(torch.randn(256, 64, generator=g(8)) / 8), h(0.1 * torch.randn(256, generator=g(9)))
    out["geglu"] = dict(x=x, weight=w, bias=b, y=R.linear_ref(x, w, b, geglu=True))
    r = h(torch.randn(48, 256, generator=g(10)))
    out["linear_add"] = dict(x=x, weight=w, bias=b, other=r, alpha=0.5, y=R.linear_ref(x, w, b, residual=r, alpha=0.5))

    x = h(torch.randn(2, 64, 9, 7, generator=g(11)))
    w, b = h(torch.randn(32, 64, 3, 3, generator=g(12)) / 24), h(0.1 * torch.randn(32, generator=g(13)))
    z = h(torch.randn(2, 32, 9, 7, generator=g(14)))
    out["conv3x3_bias_add"] = dict(x=x, weight=w, bias=b, z=z, alpha=0.5, y=R.conv2d_ref(x, w, b, z, 0.5, 1, 1))
    out["conv3x3_s2_relu"] = dict(x=x, weight=w, bias=b, y=R.conv2d_ref(x, w, b, None, 1.0, 2, 1, act="relu"))

    q = h(torch.randn(2, 70, 3, 40, generator=g(15)))
    k = h(torch.randn(2, 77, 3, 40, generator=g(16)))
    v = h(torch.randn(2, 77, 3, 40, generator=g(17)))
    out["attention_d40_kv77"] = dict(q=q, k=k, v=v, y=R.attention_ref(q, k, v))

    # inputs are fp16-exact: store them as fp16 (half the bytes); outputs stay fp32
    for case in out.values():
        for k_, v_ in list(case.items()):
            if torch.is_tensor(v_) and k_ != "y":
                case[k_] = v_.half()
    torch.save(out, os.path.join(HERE, "ops.pt"))

    # tiny UNet (SD1.5 topology): fp16-representable weights, fp32 oracle output
    m = U.build("tiny", seed=1234)
    sd = {k_: h(v_) for k_, v_ in m.state_dict().items()}
    m.load_state_dict(sd)
    sample = h(torch.randn(2, 4, 16, 16, generator=g(20)))
    ehs = h(torch.randn(2, 77, 64, generator=g(21)))
    with torch.no_grad():
        y = m(sample, 981, ehs).sample
        y2 = m(sample, torch.tensor([981.0, 1.0]), ehs).sample
    torch.save(dict(config="tiny", seed=1234, sample=sample.half(), encoder_hidden_states=ehs.half(), timestep=981, y=y,
                    timesteps_b=[981.0, 1.0], y_b=y2), os.path.join(HERE, "unet_tiny.pt"))
    # tiny VAE decoder (AutoencoderKL.decoder topology): fp16-representable weights, fp32 oracle output
    vcfg = dict(block_out_channels=(64, 128), norm_num_groups=8, layers_per_block=1)
    d = V.build("tiny", seed=4321, **vcfg)
    d.load_state_dict({k_: h(v_) for k_, v_ in d.state_dict().items()})
    zlat = h(torch.randn(2, 4, 8, 8, generator=g(30)))
    with torch.no_grad():
        yimg = d(zlat)
    torch.save(dict(config=vcfg, seed=4321, z=zlat.half(), y=yimg), os.path.join(HERE, "vae_tiny.pt"))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
