"""Sentinel token ids the splice logic keys on (values fixed by the reference's data format:
llava/constants.py there; `llava.constants` itself is NOT shadowed by the overlay and falls through
to the reference checkout)."""
IGNORE_INDEX = -100          # label value excluded from the loss
IMAGE_TOKEN_INDEX = -200     # placeholder id of the one <image> slot in input_ids
