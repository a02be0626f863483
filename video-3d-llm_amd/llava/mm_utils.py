"""llava.mm_utils surface used by the 3-D eval drivers (reference: llava/mm_utils.py:341-395).
Pure host-side string/id helpers - nothing to accelerate, kept so the drivers' imports resolve."""
import torch
from transformers import StoppingCriteria

from v3d.token_ids import IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise around '<image>' and put image_token_index in each gap (mm_utils.py:341-360)."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    ids, offset = [], 0
    if chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    for i, c in enumerate(chunks):
        if i > 0:
            ids.append(image_token_index)
        ids.extend(c[offset:])
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    if return_tensors is not None:
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria(StoppingCriteria):
    """Stop when any keyword's ids (or text) appears at the tail of the output (mm_utils.py:372-395)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        for k in keywords:
            ids = tokenizer(k).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def __call__(self, output_ids, scores, **kwargs):
        assert output_ids.shape[0] == 1, "Only support batch size 1 (yet)"
        tail = min(output_ids.shape[1] - self.start_len, 3)
        for kid in self.keyword_ids:
            kid = kid.to(output_ids.device)
            if output_ids.shape[1] >= kid.shape[0] and torch.equal(output_ids[0, -kid.shape[0]:], kid):
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -tail:], skip_special_tokens=True)[0] if tail > 0 else ""
        return any(k in text for k in self.keywords)
