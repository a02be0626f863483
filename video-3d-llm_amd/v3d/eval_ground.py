"""Entry point: `python -m v3d.eval_ground --task scanrefer|multi3drefer ...` = v3d.eval_3d (reference drivers
llava/eval/model_scanrefer.py, model_multi3drefer.py: one prefill with object proposals per question, infonce scores)."""
import sys

from .eval_3d import main

if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
