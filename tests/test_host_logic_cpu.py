"""Host-side logic that needs no GPU: config merging, the shared Adam step counter, deferred optimizer calls
(multi-rank graph dispatch), work-stream slot table."""
import math

import pytest


def test_merge_configs_is_recursive_and_never_aliases_the_defaults():
    """confignet_utils.py:39-61: nested dicts are merged key by key; the result must not share nested objects with
    the defaults (a dataset's process_metadata(config, True) rewrites config["facemodel_inputs"] in place)."""
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    before = {k: tuple(v) for k, v in DEFAULT_CONFIG["facemodel_inputs"].items()}
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "optimizer": {"lr": 1e-3}})
    assert cfg["batch_size"] == 4 and cfg["optimizer"]["lr"] == 1e-3 and cfg["optimizer"]["beta_2"] == 0.9
    assert cfg["facemodel_inputs"] is not DEFAULT_CONFIG["facemodel_inputs"]
    cfg["facemodel_inputs"]["blendshape_values"] = (62, 30)
    cfg["optimizer"]["beta_1"] = 0.5
    assert {k: tuple(v) for k, v in DEFAULT_CONFIG["facemodel_inputs"].items()} == before
    assert DEFAULT_CONFIG["optimizer"]["beta_1"] == 0.0
    extra = merge_configs(DEFAULT_CONFIG, {"use_hip_graphs": False, "facemodel_inputs": {"blendshape_values": (62, 30)}})
    assert extra["use_hip_graphs"] is False and extra["facemodel_inputs"]["blendshape_values"] == (62, 30)
    assert set(extra["facemodel_inputs"]) == set(DEFAULT_CONFIG["facemodel_inputs"])


def test_adam_shared_counter_and_lr_t():
    """[TF-2.1] R10: one iteration counter per optimizer object; lr_t = lr*sqrt(1-b2^t)/(1-b1^t)."""
    from confignet_amd import optim
    opt = optim.Adam(lr=4e-4, beta_1=0.5, beta_2=0.9)
    for t in (1, 2, 3):
        opt.iterations = t
        assert opt.lr_t() == pytest.approx(4e-4 * math.sqrt(1 - 0.9 ** t) / (1 - 0.5 ** t), rel=1e-12)
    with pytest.raises(AssertionError):
        optim.Adam(amsgrad=True)


def test_deferred_updates_record_instead_of_running():
    """Multi-rank graph dispatch: inside optim.deferred_updates() an apply_gradients call is only recorded
    (optimizer, nets, slot), in call order, and the context restores the previous state (also when nested)."""
    from confignet_amd import optim
    a, b = optim.Adam(), optim.Adam()
    n1, n2, n3 = object(), object(), object()
    with optim.deferred_updates() as outer:
        a.apply_gradients([n1, n2], advance=False, slot="g")
        with optim.deferred_updates() as inner:
            b.apply_gradients(n3, advance=False, slot="d")
        a.apply_gradients(n3, advance=False, slot="ld")
    assert [(o, nets, slot) for o, nets, slot in outer] == [(a, [n1, n2], "g"), (a, [n3], "ld")]
    assert inner == [(b, [n3], "d")]
    assert optim._deferred is None
    with pytest.raises(AssertionError):
        with optim.deferred_updates():
            a.apply_gradients(n1, advance=True)          # the counter is advanced by the host half, never inside


def test_work_stream_slots():
    """The generator step reuses the discriminator step's capture stream and the synthetic discriminator's stream for
    its second branch; the main line has its own (four streams in total)."""
    from confignet_amd.confignet_first_stage import ConfigNetFirstStage
    slots = ConfigNetFirstStage._WORK_SLOTS
    assert slots["g"] == slots["d"] and len({slots["main"], slots["d"], slots["sd"], slots["ld"]}) == 4
    assert max(slots.values()) == 3


def test_bench_launches_its_own_ranks_or_says_what_is_missing():
    """`python bench.py --gpus N` without a launcher environment starts its own N ranks; on a host with fewer devices it must
    say so and exit non-zero (it used to die on an assert about WORLD_SIZE), and a launcher environment that
    disagrees with --gpus is refused by name."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                      # (no device is visible to this check, whatever the host has)
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs 2 visible devices" in (p.stderr + p.stdout), (p.returncode, p.stderr[-500:])
    env.update(RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in (p.stderr + p.stdout), (p.returncode, p.stderr[-500:])
