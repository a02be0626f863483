"""Entry point: `python -m v3d.eval_scan2cap ...` = v3d.eval_3d with the task fixed to scan2cap (reference driver llava/eval/model_scan2cap.py)."""
import sys

from .eval_3d import main

if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:], task="scan2cap"))
