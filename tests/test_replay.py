"""Replay buffers against a fixture produced by the reference's own classes
(tests/golden/make_replay_golden.py): same evictions, same windows, same numpy RNG draws."""
import json
import os

import numpy as np
import pytest
import torch

from mapdn_amd.replay import EpisodeReplayBuffer, TransReplayBuffer

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay_golden.json")))


def _run_trans(script, device):
    buf = TransReplayBuffer(script["size"], device=device)
    for op in script["ops"]:
        if op["op"] == "add":
            ids = torch.arange(op["first"], op["first"] + op["count"], device=device)
            buf.add_experience({"id": ids, "pair": torch.stack([ids, -ids], 1).float()})
            assert len(buf) == op["len"] == len(buf.buffer)
        elif op["op"] == "get":
            np.random.seed(op["np_seed"])
            b = buf.get_batch(op["batch_size"])
            assert b["id"].tolist() == op["ids"]
            assert b["pair"][:, 1].tolist() == [-float(i) for i in op["ids"]]
        elif op["op"] == "single":
            assert int(buf.get_single(op["index"])["id"]) == op["id"]
        else:
            buf.clear()
            assert len(buf) == 0


def _run_episode(script, device):
    T = script["max_steps"]
    buf = EpisodeReplayBuffer(script["size"], T, device=device)
    for op in script["ops"]:
        if op["op"] == "add":
            lens = np.array(op["lengths"])
            ep = torch.arange(op["first"], op["first"] + len(lens), device=device)
            ids = ep[None, :] * 1000 + torch.arange(T, device=device)[:, None]           # [T, B]
            buf.add_experience({"id": ids, "x": ids.float().unsqueeze(-1)}, lens)
            assert len(buf) == op["len"]
        else:
            np.random.seed(op["np_seed"])
            b = buf.get_batch(op["batch_size"])
            assert b["id"].tolist() == op["ids"]
            assert b["x"].shape == (len(op["ids"]), 1)


@pytest.mark.parametrize("i", range(len(GOLD["trans"])))
def test_trans_buffer_matches_reference_fixture(i):
    _run_trans(GOLD["trans"][i], "cpu")


@pytest.mark.parametrize("i", range(len(GOLD["episode"])))
def test_episode_buffer_matches_reference_fixture(i):
    _run_episode(GOLD["episode"][i], "cpu")


def test_trans_buffer_errors_and_explicit_start():
    buf = TransReplayBuffer(8)
    buf.add_experience({"id": torch.arange(5)})
    with pytest.raises(ValueError):
        buf.get_batch(6)                                  # reference: np.random.choice(0, ...) raises ValueError
    with pytest.raises(KeyError):
        buf.add_experience({"other": torch.arange(2)})
    with pytest.raises(IndexError):
        buf.get_single(5)
    buf.add_experience({"id": torch.arange(5, 25)})       # more than a buffer-full at once
    assert buf.get_batch(8, start=0)["id"].tolist() == list(range(17, 25))
    with pytest.raises(IndexError):
        buf.get_batch(4, start=5)
    with pytest.raises(ValueError):
        TransReplayBuffer(0)


@pytest.mark.gpu
def test_replay_buffers_on_device():
    for s in GOLD["trans"]:
        _run_trans(s, "cuda:0")
    for s in GOLD["episode"]:
        _run_episode(s, "cuda:0")
    buf = TransReplayBuffer(1 << 16, device="cuda:0")
    x = torch.randn(4096, 22, 58, device="cuda:0")
    for _ in range(20):
        buf.add_experience({"state": x, "done": torch.zeros(4096, 1, dtype=torch.bool, device="cuda:0")})
    b = buf.get_batch(1024)
    assert b["state"].is_cuda and b["state"].shape == (1024, 22, 58) and len(buf) == 1 << 16


def test_non_wrapping_window_is_a_view_of_the_ring():
    """round 5: a sampled window that does not wrap is handed out as views (no gathered copy: 7 % of the end-to-end loop at
    32 x 8192 transitions); a wrapping one as a gathered copy — same values either way"""
    import torch
    from mapdn_amd.replay import TransReplayBuffer
    buf = TransReplayBuffer(10)
    for t in range(13):                                   # the ring has wrapped: oldest entry is t = 3 at ring position 3
        buf.add_experience({"x": torch.full((1, 2), float(t)), "i": torch.tensor([t])})
    assert len(buf) == 10
    a = buf.get_batch(4, start=1)                         # entries t = 4..7: ring positions 4..7, no wrap
    assert a["i"].tolist() == [4, 5, 6, 7] and a["x"].data_ptr() == buf.store["x"][4].data_ptr()
    b = buf.get_batch(4, start=5)                         # entries t = 8..11: ring positions 8, 9, 0, 1 — wraps
    assert b["i"].tolist() == [8, 9, 10, 11] and b["x"][:, 0].tolist() == [8.0, 9.0, 10.0, 11.0]
    lo, hi = buf.store["x"].data_ptr(), buf.store["x"].data_ptr() + buf.store["x"].numel() * 4
    assert not (lo <= b["x"].data_ptr() < hi)


def test_mirrored_ring_makes_every_window_a_view():
    """TransReplayBuffer(size, window=w): the first w ring positions are mirrored behind the ring, so a window that wraps is still one
    slice.  Same contents as the plain ring for every start, through several wrap-arounds and multi-transition insertions."""
    import torch
    from mapdn_amd.replay import TransReplayBuffer
    rng = np.random.default_rng(0)
    plain, mir = TransReplayBuffer(12), TransReplayBuffer(12, window=5)
    nxt = 0
    for step in range(40):
        b = int(rng.integers(1, 5))
        t = {"x": torch.arange(nxt, nxt + b, dtype=torch.float32).view(b, 1).repeat(1, 3), "i": torch.arange(nxt, nxt + b)}
        nxt += b
        plain.add_experience(t); mir.add_experience({k: v.clone() for k, v in t.items()})
        assert len(plain) == len(mir)
        for w in (1, 3, 5):
            if len(mir) < w:
                continue
            for start in range(len(mir) - w + 1):
                a, m = plain.get_batch(w, start=start), mir.get_batch(w, start=start)
                assert torch.equal(a["i"], m["i"]) and torch.equal(a["x"], m["x"])
                lo = mir.store["x"].data_ptr()
                assert lo <= m["x"].data_ptr() < lo + mir.store["x"].numel() * 4        # always a view
    big = mir.get_batch(7, start=0)                       # wider than the mirror: still correct (copied when it wraps)
    assert torch.equal(big["i"], plain.get_batch(7, start=0)["i"])
