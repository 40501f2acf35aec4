"""Same command line as the reference's Hydra entry (main.py:15-35):

    python main.py --config-name=delete_celeb [--config-path=config] [key=value ...]

Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N main.py --config-name=...``.
"""
import argparse
import datetime
import os
import sys
import uuid

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from siss_amd import hydra_lite  # noqa: E402
from siss_amd.tasks import Task  # noqa: E402,F401


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-name", required=True)
    ap.add_argument("--config-path", default=os.path.join(ROOT, "config"))
    args, overrides = ap.parse_known_args(argv)
    cfg = hydra_lite.compose(args.config_name, args.config_path, overrides)
    if cfg.get("resume_from_checkpoint"):
        cfg.output_dir = f"{cfg.output_dir}/{os.path.dirname(cfg.resume_from_checkpoint)}"
    else:
        try:
            from zoneinfo import ZoneInfo
            now = datetime.datetime.now(tz=ZoneInfo("US/Pacific"))
        except Exception:
            now = datetime.datetime.now()
        cfg.output_dir = f"{cfg.output_dir}/{now.strftime('%Y-%m-%d_%H-%M-%S')}_{uuid.uuid4()}"
    task = hydra_lite.instantiate(cfg.task, cfg=cfg, _recursive_=False)
    task.run()


if __name__ == "__main__":
    main()
