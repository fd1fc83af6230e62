"""Prompt assembly for the streaming agent: the reference's `preprocess_qwen` + conversation template.

Mirrors streamvln/streamvln_eval.py:290-304,393-469 (Habitat loop: random conjunction, "These are your historical
observations <memory>.") and streamvln/streamvln_agent.py:109-167,201-215 (real-world agent: fixed conjunction,
"You have visited these areas <memory>.").  Needs a HF tokenizer (not available offline; tests use a stub with the
same methods).  Returns a python list of ids with the sentinels -200 / -300 in place of <image> / <memory>.
"""
from __future__ import annotations

import copy
import random
from typing import List, Optional

from .agent import CONJUNCTIONS
from .config import (DEFAULT_IMAGE_TOKEN, DEFAULT_MEMORY_TOKEN, DEFAULT_VIDEO_TOKEN, IMAGE_TOKEN_INDEX, MEMORY_TOKEN_INDEX)

# streamvln_agent.py:48 / streamvln_eval.py (same text)
PROMPT = ("<video>\nYou are an autonomous navigation assistant. Your task is to <instruction>. Devise an action sequence to follow "
          "the instruction using the four actions: TURN LEFT (←) or TURN RIGHT (→) by 15 degrees, MOVE FORWARD (↑) by 25 "
          "centimeters, or STOP.")
CHAT_TEMPLATE = ("{% for message in messages %}{{'<|im_start|>' + message['role'] + '\n' + message['content'] + '<|im_end|>' + '\n'}}"
                 "{% endfor %}{% if add_generation_prompt %}{{ '<|im_start|>assistant\n' }}{% endif %}")


class QwenPromptEncoder:
    """Callable(first_turn, with_memory, instruction) -> ids, the `prompt_encoder` of StreamingAgent."""

    def __init__(self, tokenizer, flavour: str = "eval", system_message: str = "You are a helpful assistant.",
                 rng: Optional[random.Random] = None):
        assert flavour in ("eval", "agent")
        self.flavour = flavour
        self.system_message = system_message
        self.rng = rng or random
        tok = copy.deepcopy(tokenizer)                                   # streamvln_eval.py:399-403
        tok.add_tokens(["<image>"], special_tokens=True)
        tok.add_tokens(["<memory>"], special_tokens=True)
        tok.chat_template = CHAT_TEMPLATE                                # :413-414 (no implicit system message)
        self.tok = tok
        self.image_id = tok.convert_tokens_to_ids("<image>")
        self.memory_id = tok.convert_tokens_to_ids("<memory>")

    def first_turn_text(self, with_memory: bool, instruction: str) -> str:
        text = PROMPT
        if with_memory:                                                  # step_id != 0
            text += (f" These are your historical observations {DEFAULT_MEMORY_TOKEN}." if self.flavour == "eval"
                     else f" You have visited these areas {DEFAULT_MEMORY_TOKEN}.")
        text = text.replace(DEFAULT_VIDEO_TOKEN + "\n", "")
        return text.replace("<instruction>.", instruction)

    def __call__(self, first_turn: bool, with_memory: bool, instruction: str = "") -> List[int]:
        conj = self.rng.choice(CONJUNCTIONS) if self.flavour == "eval" else CONJUNCTIONS[0]     # eval:424 / agent:126
        prompt = conj + DEFAULT_IMAGE_TOKEN
        human = self.first_turn_text(with_memory, instruction) + f" {prompt}." if first_turn else f"{prompt}."
        ids: List[int] = []
        if first_turn:                                                   # add_system=True only on the first turn of a window
            ids += self.tok.apply_chat_template([{"role": "system", "content": self.system_message}])
        for role, content in (("user", human), ("assistant", "")):
            ids += self.tok.apply_chat_template([{"role": role, "content": content}])
        return [IMAGE_TOKEN_INDEX if t == self.image_id else MEMORY_TOKEN_INDEX if t == self.memory_id else int(t) for t in ids]
