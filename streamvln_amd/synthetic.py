"""Synthetic streams for parity and perf runs.

No tokenizer, dataset or checkpoint exists offline, so runs use
  * frames: uint8 [480,640,3] from numpy.random.default_rng(1000 + env*10007 + step)
  * token ids: uniform text ids with `<image>` (-200) / `<memory>` (-300) sentinels at fixed
    offsets; first turn 181 ids, memory first turn 190 ids, later turns 16 ids.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .config import IMAGE_TOKEN_INDEX, MEMORY_TOKEN_INDEX, StreamVLNConfig

FRAME_H, FRAME_W = 480, 640


def synthetic_frame(env: int, step: int) -> np.ndarray:
    rng = np.random.default_rng(1000 + env * 10007 + step)
    return rng.integers(0, 256, size=(FRAME_H, FRAME_W, 3), dtype=np.uint8)


class SyntheticPromptEncoder:
    """Stands in for `preprocess_qwen` + tokenizer (streamvln_eval.py:393-469): returns ids
    of the lengths the real prompts have, with sentinels where the real prompt has them."""

    def __init__(self, cfg: StreamVLNConfig, seed: int = 7, first_len: int = 181, memory_len: int = 190,
                 later_len: int = 16):
        self.cfg = cfg
        self.seed = seed
        self.first_len, self.memory_len, self.later_len = first_len, memory_len, later_len
        self._n = 0

    def reset(self):
        """start the prompt stream over: two runs that should see the same prompts (an A/B of two engine modes) each begin here"""
        self._n = 0

    def _text(self, n: int) -> List[int]:
        lo = min(1000, self.cfg.vocab // 4)
        hi = min(150000, self.cfg.vocab)
        rng = np.random.default_rng(self.seed + 7919 * self._n)
        self._n += 1
        return [int(t) for t in rng.integers(lo, hi, size=n)]

    def __call__(self, first_turn: bool, with_memory: bool, instruction: str = "") -> List[int]:
        if not first_turn:
            ids = self._text(self.later_len - 1)
            ids.insert(self.later_len - 3, IMAGE_TOKEN_INDEX)      # "... <image>.<|im_end|>\n"
            return ids
        if with_memory:
            ids = self._text(self.memory_len - 2)
            ids.insert(self.memory_len - 12, MEMORY_TOKEN_INDEX)   # "... observations <memory>. you can see <image>."
            ids.insert(self.memory_len - 3, IMAGE_TOKEN_INDEX)
            return ids
        ids = self._text(self.first_len - 1)
        ids.insert(self.first_len - 3, IMAGE_TOKEN_INDEX)
        return ids
