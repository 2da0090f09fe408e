"""Two ranks through the ENTRY POINT (not a pytest file: started under torch.distributed.run by tests/test_parallel_gpu.py).
`xmc_gan/train_gan.py`'s `main()` on every rank -- process group, parameter broadcast, per-rank synthetic batches, `train()` replaying graph
segments with the collectives as seams -- then each rank writes its final weights and last losses for the test to compare."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    import xmc_gan.train_gan as tg
    last = tg.main(["--cfg", a.cfg, "--synthetic", str(a.steps), "--max_epoch", "1", "--precision", "fp32", "--seed", "21",
                    "--output_dir", os.path.join(a.out, "run")])
    netG, netD = tg.main.last_models
    w = {"G." + k: v.detach().float().cpu() for k, v in netG.state_dict().items()}
    w.update({"D." + k: v.detach().float().cpu() for k, v in netD.state_dict().items()})
    losses = {k: float(v) for k, v in last.items() if torch.is_tensor(v) and v.numel() == 1}
    torch.save({"weights": w, "losses": losses, "hipgraph": bool(last.get("hipgraph", False))}, os.path.join(a.out, f"rank{rank}.pt"))


if __name__ == "__main__":
    main()
