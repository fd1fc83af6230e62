"""Synthetic streaming scenarios shared by the golden generator (oracle/make_golden.py) and the
parity tests (test infrastructure)."""
import torch

from streamvln_amd.agent import StreamingAgent
from streamvln_amd.config import TINY, TRUE1, TRUE4
from streamvln_amd.synthetic import SyntheticPromptEncoder, synthetic_frame

SEED = 1234

SCENARIOS = {
    # 36 env steps = 9 model turns over 3 windows; EOS set = ids % 3 == 2 (varying turn lengths, cap 6)
    "tiny_episode": dict(cfg=TINY, steps=36, num_frames=12, nfs=4, num_history=2, max_new=6, eos_mod=3,
                         lens=(40, 48, 16)),
    # true dimensions, one ViT layer + one LLM layer, 36 env steps = 9 turns: first turn (T=376), seven steady turns (T=214) and the
    # window restart at step 32 (9 views: 8 history frames -> 1568-row <memory> block + current frame, T=1952), 3 tokens each
    "true1_episode": dict(cfg=TRUE1, steps=36, num_frames=32, nfs=4, num_history=8, max_new=3, eos_mod=0,
                          lens=(181, 190, 16)),
    # true dimensions, 4 ViT + 4 LLM layers, full vocabulary (152 064): first turn + one steady turn, 4 tokens each
    "true4_episode": dict(cfg=TRUE4, steps=8, num_frames=32, nfs=4, num_history=8, max_new=4, eos_mod=0,
                          lens=(181, 190, 16)),
    # config.tokenizer_model_max_length = 150 (stream_video_vln.py:241-244): every turn's spliced rows (235 first, 211 later) are cut to 150,
    # in the middle of the image block
    "tiny_truncate": dict(cfg=TINY, steps=12, num_frames=12, nfs=4, num_history=2, max_new=4, eos_mod=0, lens=(40, 48, 16), tml=150),
    # generation_config.repetition_penalty = 1.3 (transformers RepetitionPenaltyLogitsProcessor under greedy decoding): the unpenalised
    # tiny run repeats ids inside a turn ([1507, 153, 153, 153, ...]), so the penalty changes the sequence
    "tiny_penalty": dict(cfg=TINY, steps=16, num_frames=12, nfs=4, num_history=2, max_new=6, eos_mod=0, lens=(40, 48, 16), rep_penalty=1.3),
}


def apply_knobs(model, sc):
    """the two call-time attributes of the reference's model object, set the same way on the reference, the oracle and the HIP model"""
    if "tml" in sc:
        model.config.tokenizer_model_max_length = sc["tml"]
    if "rep_penalty" in sc:
        model.generation_config.repetition_penalty = sc["rep_penalty"]


def eos_ids(sc):
    return tuple(range(sc["eos_mod"] - 1, sc["cfg"].vocab, sc["eos_mod"])) if sc["eos_mod"] else ()


def run_scenario(model, sc, preprocess, on_turn=None, device="cpu", image_dtype=torch.float32, steps=None):
    cfg = sc["cfg"]
    enc = SyntheticPromptEncoder(cfg, seed=sc.get("prompt_seed", 7), first_len=sc["lens"][0], memory_len=sc["lens"][1], later_len=sc["lens"][2])
    agent = StreamingAgent(model, enc, num_frames=sc["num_frames"], num_future_steps=sc["nfs"],
                           num_history=sc["num_history"], max_new_tokens=sc["max_new"], eos_token_ids=eos_ids(sc),
                           preprocess=preprocess, device=device, image_dtype=image_dtype)
    seen = 0
    for step in range(steps or sc["steps"]):
        agent.act(synthetic_frame(0, step))
        if on_turn is not None and len(agent.turn_log) > seen:
            seen = len(agent.turn_log)
            on_turn(seen - 1, agent.turn_log[-1])
    return agent.turn_log
