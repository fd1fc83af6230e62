"""Deterministic stand-in for the Qwen2 tokenizer (none exists offline): the methods `preprocess_qwen` and the
agent loop use (streamvln/streamvln_agent.py:109-167,250; streamvln/streamvln_eval.py:393-469).  Test infrastructure:
shared by oracle/make_agent_calls.py (which drives the REFERENCE's VLNEvaluator with it) and the host-logic tests
(which drive this project's StreamingAgent / QwenPromptEncoder with it), so both sides tokenize identically.

ids: fixed ids for the specials, 1000 + crc32(word) % 100000 for everything else (order independent)."""
import re
import zlib

SPECIAL = {"<|endoftext|>": 151643, "<|im_start|>": 151644, "<|im_end|>": 151645, "\n": 198}
ACTION_WORDS = {11: "↑", 12: "←", 13: "→", 10: "STOP"}
_SPLIT = re.compile(r"(<\|im_start\|>|<\|im_end\|>|<\|endoftext\|>|<image>|<memory>|\n|[A-Za-z0-9_']+|[^\sA-Za-z0-9_'])")


class StubTokenizer:
    def __init__(self):
        self.added = {}
        self.chat_template = None

    # -- vocabulary ----------------------------------------------------------------------------------
    def _id(self, w):
        if w in SPECIAL:
            return SPECIAL[w]
        if w in self.added:
            return self.added[w]
        return 1000 + zlib.crc32(w.encode("utf-8")) % 100000

    def add_tokens(self, toks, special_tokens=False):
        for t in toks:
            self.added.setdefault(t, 151646 + len(self.added))
        return len(toks)

    def convert_tokens_to_ids(self, t):
        return self._id(t)

    @property
    def additional_special_tokens_ids(self):
        return [SPECIAL["<|im_start|>"], SPECIAL["<|im_end|>"]]

    def encode(self, text):
        return [self._id(w) for w in _SPLIT.findall(text)]

    def __call__(self, text):
        return type("Enc", (), {"input_ids": self.encode(text)})()

    # -- the chat template both callers install (streamvln_agent.py:117, streamvln_eval.py:413) -------
    def apply_chat_template(self, msgs, add_generation_prompt=False):
        assert self.chat_template is not None and "<|im_start|>" in self.chat_template
        text = "".join("<|im_start|>" + m["role"] + "\n" + m["content"] + "<|im_end|>" + "\n" for m in msgs)
        if add_generation_prompt:
            text += "<|im_start|>assistant\n"
        return self.encode(text)

    def batch_decode(self, ids, skip_special_tokens=False):
        return ["".join(ACTION_WORDS.get(int(t), "") for t in row) + " " for row in ids]


class RecordingModel:
    """Model stand-in behind the reference's operator surface: records every `generate(**kwargs)` and every
    `reset_for_env`; answers with a call-indexed token sequence (1..4 action tokens + <|im_end|>)."""

    def __init__(self, image_processor):
        from types import SimpleNamespace
        self._tower = SimpleNamespace(image_processor=image_processor)
        self.calls, self.resets = [], []

    def get_vision_tower(self):
        return self._tower

    def reset_for_env(self, idx):
        self.resets.append((len(self.calls), int(idx)))

    def generate(self, **kw):
        import torch
        self.calls.append(summarize_call(kw))       # at call time: the reference passes its live `time_ids` list
        n = len(self.calls)
        toks = [11 + (n + j) % 3 for j in range(1 + n % 4)] + [SPECIAL["<|im_end|>"]]
        out = type("Out", (), {})()
        out.sequences = torch.tensor([toks], dtype=torch.long)
        out.past_key_values = ("kv", n)
        return out


def summarize_call(kw):
    """The caller-visible content of one generate call, as arrays (what tests/golden/agent_calls.npz stores)."""
    import numpy as np
    import torch
    img = kw["images"]
    pkv = kw.get("past_key_values")
    return {
        "inputs": kw["inputs"].reshape(-1).cpu().numpy().astype(np.int64),
        "views": np.int64(img.shape[1]),
        "image_shape": np.asarray(img.shape, dtype=np.int64),
        "image_is_bf16": np.int64(img.dtype == torch.bfloat16),
        "view_sums": img[0].to(torch.float64).sum(dim=(1, 2, 3)).cpu().numpy(),
        "time_ids": np.asarray(kw["time_ids"][0], dtype=np.int64),
        "env_id": np.int64(kw["env_id"]),
        "pkv": np.int64(-1 if pkv is None else pkv[1]),
        "flags": np.asarray([int(kw["do_sample"]), int(kw["num_beams"]), int(kw["max_new_tokens"]), int(kw["use_cache"]),
                             int(kw["return_dict_in_generate"])], dtype=np.int64),
    }
