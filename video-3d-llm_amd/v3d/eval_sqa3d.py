"""Entry point: `python -m v3d.eval_sqa3d ...` = v3d.eval_3d with the task fixed to sqa3d (reference driver llava/eval/model_sqa3d.py)."""
import sys

from .eval_3d import main

if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:], task="sqa3d"))
