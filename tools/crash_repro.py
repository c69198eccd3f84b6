"""Amplified reproducer for the intermittent crash of the compiled ControlNet -> UNet chain (rounds 4 / 5).

One process repeats the sequence the failing test runs once -- build a tiny ControlNet and a tiny UNet, compile both with graphs on,
run the chain twice, drop everything -- `--iters` times, optionally after creating and destroying an RCCL process group (what
`test_rccl_weight_broadcast_single_rank` used to do inside the pytest process). Test infrastructure: imports `oracle` for the module
builders, as tests/ do. Product A/B knobs are environment variables read by sfast.engine.unet2d:
  SFAST_FORK_EVENTS_LOCAL=1     fork / join events of the forked capture die inside the capture (round-4 behaviour)
  SFAST_GRAPH_DESTROY_LOSER=1   the graph that loses the serial-vs-forked calibration is destroyed right away (round-4 behaviour)
Run under tools/_crashbt.so (LD_PRELOAD) to get the C backtrace of the dying thread.
"""
import argparse
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stable-fast_amd"), os.path.join(ROOT, "tests")]

import torch  # noqa: E402


def rccl_round_trip():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(1 << 20, device="cuda")
        dist.broadcast(t, src=0)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rccl", type=int, default=0)
    ap.add_argument("--graphs", type=int, default=1)
    ap.add_argument("--budget", type=float, default=1e9, help="stop starting new iterations after this many seconds")
    a = ap.parse_args()
    from oracle import controlnet_ref as CN
    from oracle import unet_ref as U
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    t0 = time.time()
    if a.rccl:
        rccl_round_trip()
        print("rccl round trip done", flush=True)
    ccfg, ucfg = CN.tiny_config(), U.tiny_config()
    g = torch.Generator().manual_seed(9)
    hw = ucfg["sample_size"]
    sample = torch.randn(2, ucfg["in_channels"], hw, hw, generator=g).to("cuda", torch.float16)
    ehs = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g).to("cuda", torch.float16)
    cond = torch.rand(2, 3, 64, 64, generator=g).to("cuda", torch.float16)
    first = None
    for it in range(a.iters):
        if time.time() - t0 > a.budget:
            break
        cnet = CN.build(ccfg, seed=41, dtype=torch.float16, device="cuda")
        unet = U.build(ucfg, seed=42, dtype=torch.float16, device="cuda")
        c = CompilationConfig.Default()
        c.enable_cuda_graph = bool(a.graphs)
        cnet = compile_unet(cnet, c)
        unet = compile_unet(unet, c)
        for _ in range(2):
            out = cnet(sample, 444, encoder_hidden_states=ehs, controlnet_cond=cond, return_dict=True)
            y = unet(sample, 444, encoder_hidden_states=ehs, down_block_additional_residuals=out.down_block_res_samples,
                     mid_block_additional_residual=out.mid_block_res_sample, return_dict=False)[0]
        torch.cuda.synchronize()
        s = float(y.float().abs().sum())
        first = s if first is None else first
        if s != first:
            print(f"NOTE: iteration {it} checksum {s} != first {first}", flush=True)
        del cnet, unet, out, y
        gc.collect()
        print(f"iter {it} ok {time.time() - t0:.1f}s", flush=True)
    print("REPRO_DONE", flush=True)


if __name__ == "__main__":
    main()
