"""Host-side streaming agent: the caller of the hot path.

Mirrors the reference's `VLNEvaluator.step` (streamvln/streamvln_agent.py:169-258) and the
identical turn logic inlined in the Habitat loop (streamvln/streamvln_eval.py:290-350):

  * every env step records the preprocessed frame and its time id;
  * a model turn happens when the pending action queue is empty: first turn of a window
    builds the system+instruction prompt (with `<memory>` iff step_id != 0), later turns a
    short "<conjunction> <image>." prompt; `inputs = cat(prev output_ids, new ids)`;
  * at `step_id % num_frames == 0` (step_id != 0) the history frames
    `rgb_list[0 : t0 : t0 // num_history]` are prepended (streamvln_eval.py:313-321);
  * after the step that makes `step_id % num_frames == 0` the window is reset:
    `model.reset_for_env`, `output_ids = None`, `past_key_values = None`, `time_ids = []`
    (streamvln_eval.py:346-350).

The model is anything exposing the reference's operator surface
(`generate(...)`, `reset_for_env(i)`, `get_vision_tower().image_processor`): the HIP-backed
`streamvln_amd.StreamVLNForCausalLM`, or the CPU oracle in tests.
"""
from __future__ import annotations

import itertools
import re
from collections import OrderedDict
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

ACTIONS2IDX = OrderedDict({"STOP": [0], "↑": [1], "←": [2], "→": [3]})   # streamvln_agent.py:52-57

CONJUNCTIONS = [                                                        # streamvln_agent.py:60-68
    "you can see ", "in front of you is ", "there is ", "you can spot ",
    "you are toward the ", "ahead of you is ", "in your sight is ",
]


def parse_actions(output: str) -> List[int]:
    """streamvln_agent.py:101-107 / streamvln_eval.py:382-389."""
    pattern = "|".join(re.escape(a) for a in ACTIONS2IDX)
    matches = re.compile(pattern).findall(output)
    return list(itertools.chain.from_iterable(ACTIONS2IDX[m] for m in matches))


class StreamingAgent:
    """Stateful single-env agent.  `prompt_encoder(first_turn, with_memory, instruction)`
    returns the new turn's token ids (with -200/-300 sentinels); `decode_actions(ids)` turns the
    generated ids into an action list (tokenizer.batch_decode + parse_actions in the reference)."""

    def __init__(self, model, prompt_encoder: Callable, num_frames: int = 32, num_future_steps: int = 4,
                 num_history: Optional[int] = 8, env_id: int = 0, device: str = "cpu",
                 image_dtype: torch.dtype = torch.float32, max_new_tokens: int = 10000,
                 eos_token_ids: Sequence[int] = (), decode_actions: Optional[Callable] = None,
                 preprocess: Optional[Callable] = None, ids_device: Optional[str] = "cpu"):
        self.model = model
        self.prompt_encoder = prompt_encoder
        self.num_frames, self.num_future_steps, self.num_history = num_frames, num_future_steps, num_history
        self.env_id, self.device, self.image_dtype = env_id, device, image_dtype
        # the reference callers put `inputs` on the model's device (streamvln_eval.py:326); the engine takes token ids from the host, so
        # this agent leaves them there (no H2D + D2H of a few dozen ids per turn; `sequences` come back on the same device).  None = as
        # the reference: `device`.  The model accepts both.
        self.ids_device = device if ids_device is None else ids_device
        self.max_new_tokens, self.eos_token_ids = max_new_tokens, tuple(eos_token_ids)
        self.decode_actions = decode_actions or (lambda ids: [1] * num_future_steps)
        if preprocess is None:
            proc = model.get_vision_tower().image_processor
            preprocess = lambda rgb: proc.preprocess_array(rgb)
        self.preprocess = preprocess
        self.turn_log: List[dict] = []
        self._aux: dict = {}
        self.pending_turn = False          # AsyncBatchedAgents: the model turn of the current env step has returned, the step itself is pending
        self.reset_memory()

    def reset_memory(self):                                   # streamvln_agent.py:87-99
        self.rgb_list: List[torch.Tensor] = []
        self.time_ids: List[int] = []
        self.action_seq: List[int] = []
        self.output_ids = None
        self.past_key_values = None
        self.step_id = 0
        self.last_image = None
        self.model.reset_for_env(self.env_id)

    # -- one model turn ---------------------------------------------------------------
    def _build_request(self, instruction: str, env_id: Optional[int] = None) -> dict:
        env_id = self.env_id if env_id is None else env_id
        first = self.output_ids is None
        with_memory = first and self.step_id != 0
        ids = torch.tensor([self.prompt_encoder(first, with_memory, instruction)], dtype=torch.long)
        if not first:
            ids = torch.cat([self.output_ids.to(ids.device), ids], dim=1)
        images = self.rgb_list[-1:]
        if self.step_id != 0 and self.step_id % self.num_frames == 0:
            t0 = self.time_ids[0]
            if self.num_history is None:
                hist = slice(0, t0, self.num_future_steps)
            else:
                hist = slice(0, t0, t0 // self.num_history)
            images = self.rgb_list[hist] + images
        V = len(images)
        self._pending = {"step_id": self.step_id, "n_inputs": int(ids.shape[1]), "views": V, "memory": bool(with_memory)}
        # torch.stack(images) of the reference (streamvln_eval.py:313-321); one view: the same values as a view of the frame (no copy kernel)
        self._prime_stack(images[0])
        stacked = images[0].unsqueeze(0) if V == 1 else torch.stack(images)
        aux = self._aux.get(V)
        if aux is None:              # depths / poses / intrinsics are built by the reference callers and ignored by the model
            aux = self._aux[V] = (torch.zeros(1, V, 1, 1), torch.zeros(1, V, 4, 4), torch.zeros(1, V, 4, 4))
        return {
            "images": stacked.unsqueeze(0).to(self.device).to(self.image_dtype),
            "depths": aux[0], "poses": aux[1], "intrinsics": aux[2],
            "inputs": ids.to(self.ids_device), "env_id": env_id, "time_ids": [list(self.time_ids)], "task_type": [0],
            "do_sample": False, "num_beams": 1, "max_new_tokens": self.max_new_tokens, "use_cache": True,
            "return_dict_in_generate": True, "past_key_values": self.past_key_values, "eos_token_ids": self.eos_token_ids,
        }

    def _prime_stack(self, frame):
        """torch loads the copy kernel of an N-input torch.stack at its first use: the 9-view stack of the first window restart of a process
        took 8.5-18 ms (0.1 ms at every later restart).  The first turn of an agent therefore runs that stack once on its own frame, so
        the one-time cost sits at the start of the first episode and not in the first restart turn."""
        if not getattr(self, "_stack_primed", False):
            self._stack_primed = True
            if frame.is_cuda and self.num_history:
                torch.stack([frame] * (self.num_history + 1))

    def _consume(self, out):
        self.output_ids = out.sequences
        self.past_key_values = out.past_key_values
        self.turn_log.append(dict(self._pending, out=out))
        actions = list(self.decode_actions(out.sequences))
        return actions if len(actions) else [0]               # streamvln_eval.py:340-341

    def _turn(self, instruction: str, env_id: Optional[int] = None):
        return self._consume(self.model.generate(**self._build_request(instruction, env_id)))

    # -- split form of act() used by BatchedAgents: observe -> (maybe) request -> finish ----------
    def observe(self, rgb):
        self.time_ids.append(self.step_id)
        self.rgb_list.append(self.preprocess(rgb))
        return len(self.action_seq) == 0                      # True: this env needs a model turn now

    def finish_step(self) -> int:
        action = self.action_seq.pop(0)
        self.step_id += 1
        if self.step_id % self.num_frames == 0:
            self.model.reset_for_env(self.env_id)
            self.output_ids = None
            self.past_key_values = None
            self.time_ids = []
        return action

    # -- eval-loop flavour: one env step, returns the action taken -------------------
    def act(self, rgb: np.ndarray, instruction: str = "") -> int:
        """One environment step of the Habitat loop (streamvln_eval.py:247-350)."""
        if self.observe(rgb):
            self.action_seq = self._turn(instruction)
        return self.finish_step()

    # -- real-world flavour ------------------------------------------------------------
    def step(self, idx: int, rgb: np.ndarray, instruction_text: str = "", run_model: bool = False):
        """`VLNEvaluator.step` (streamvln_agent.py:169-258).  The caller increments
        `self.step_id` (http_realworld_server.py:112)."""
        if run_model:
            self.last_image = self.preprocess(rgb)
        image = self.last_image
        self.time_ids.append(self.step_id)
        self.rgb_list.append(image)
        if not run_model:
            if (self.step_id + 1) % self.num_frames == 0:
                self.model.reset_for_env(idx)
                self.output_ids = None
                self.past_key_values = None
                self.time_ids = []
            return None, 0, None
        actions = self._turn(instruction_text, idx)
        return actions, 0.0, self.turn_log[-1]["out"]


class BatchedAgents:
    """Several envs on one GPU stepped in lockstep (BASELINE configs[4]: DAgger-style concurrent envs; the reference
    runs one env per process, streamvln_dagger.py:299).  Each env keeps its own StreamingAgent state; the turns that fall
    due at the same step are sent to the model together (`generate_batch`)."""

    def __init__(self, agents):
        self.agents = list(agents)
        self.model = self.agents[0].model

    def act(self, rgbs, instructions=None):
        instructions = instructions or [""] * len(self.agents)
        due = [a for a, rgb in zip(self.agents, rgbs) if a.observe(rgb)]
        if due:
            reqs = [a._build_request(instructions[self.agents.index(a)]) for a in due]
            shared = {k: reqs[0][k] for k in ("max_new_tokens", "eos_token_ids")}
            outs = self.model.generate_batch(reqs, **shared)
            for a, out in zip(due, outs):
                a.action_seq = a._consume(out)
        return [a.finish_step() for a in self.agents]


class AsyncBatchedAgents:
    """Several envs on one GPU whose model turns fall due at DIFFERENT times (the DAgger collector's situation: each env mixes
    expert steps, which need no model call, with model steps; streamvln_dagger.py:232-313).  `tick(rgbs)` advances every env that
    is not waiting for the model by one env step (an env whose action queue is empty submits its turn instead) and then runs ONE
    scheduler iteration (`model.step_batch`): envs that submitted in earlier ticks are decoding while new ones prefill, all in the
    same pass over the weights.  Returns {agent index: action taken} for the envs that stepped in this tick."""

    def __init__(self, agents, on_result=None):
        self.agents = list(agents)
        self.model = self.agents[0].model
        self.waiting = {}                                 # scheduler slot -> agent index
        self.on_result = on_result                        # optional callback(agent index, ticket, GenerateOutput)
        self.stats = {"iterations": 0, "mixed_iterations": 0, "max_in_flight": 0}

    def tick(self, rgbs, instructions=None, active=None):
        """`rgbs[i]`: the frame of agent i's current env step (ignored for agents that are waiting); `active`: indices allowed to
        step in this tick (default: all)."""
        instructions = instructions or [""] * len(self.agents)
        acted = {}
        busy = set(self.waiting.values())
        for i, a in enumerate(self.agents):
            if i in busy or (active is not None and i not in active):
                continue
            if a.pending_turn:                            # its turn came back in an earlier tick: take the env step now
                a.pending_turn = False
                acted[i] = a.finish_step()
                continue
            if a.observe(rgbs[i]):
                t = self.model.submit(**a._build_request(instructions[i]))
                self.waiting[t.slot] = i
            else:
                acted[i] = a.finish_step()
        if not self.waiting:
            return acted
        n_new = len(self.waiting) - len(busy)
        self.stats["iterations"] += 1
        self.stats["mixed_iterations"] += int(n_new > 0 and len(busy) > 0)      # new turns prefill while older ones decode
        self.stats["max_in_flight"] = max(self.stats["max_in_flight"], len(self.waiting))
        done, _ = self.model.step_batch()
        for ticket, out in done:
            i = self.waiting.pop(ticket.slot)
            a = self.agents[i]
            a.action_seq = a._consume(out)
            a.pending_turn = True
            if self.on_result is not None:
                self.on_result(i, ticket, out)
        return acted
