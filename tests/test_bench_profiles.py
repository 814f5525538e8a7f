"""bench.py's roofline inputs: a committed PMC profile is used only for a run with the same
workload, viewport, view, schedule and kernel sources; anything else is reported as stale, never silently used."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _latest_round():
    """profiles/r<N> with the largest N that holds PMC summaries: the round whose kernels the tree holds."""
    import glob
    import re
    rounds = sorted({int(re.search(r"r(\d+)$", os.path.dirname(p)).group(1))
                     for p in glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_issue*.json"))})
    return "r%d" % rounds[-1]


def _committed(kind):
    import glob
    out = []
    for path in glob.glob(os.path.join(ROOT, "profiles", _latest_round(), "pmc_%s*.json" % kind)):
        with open(path) as f:
            for entry in json.load(f).values():
                out.append((path, entry))
    return out


def test_profiles_describe_themselves():
    for kind in ("issue", "traffic"):
        entries = _committed(kind)
        assert entries, kind
        for path, e in entries:
            m = e.get("meta")
            assert m, path
            for k in ("workload", "viewport", "view", "frames_in_flight", "frames_per_launch", "round_budget",
                      "source_hash", "steps", "warmup"):
                assert k in m, (path, k)
            assert len(m["source_hash"].split("+")[0]) == 16


def test_a_profile_is_used_only_for_its_own_schedule_and_sources():
    path, e = next((p, e) for p, e in _committed("issue") if e["meta"]["workload"] == "shells2048"
                   and e["meta"]["frames_per_launch"] >= 16)   # (the default throughput schedule's profile)
    m = e["meta"]
    key = bench.schedule_key(m["workload"], m["viewport"], m["view"], m["frames_in_flight"], m["frames_per_launch"],
                             m["round_budget"])
    hit, stale = bench.find_profile("issue", key, m["source_hash"])
    assert stale is None and hit is not None and hit["valu_wave_insts_per_frame"] == e["valu_wave_insts_per_frame"]
    assert hit["file"].startswith("profiles/%s/" % _latest_round())
    # other kernel sources: nothing is used, the nearest profile is named
    hit, stale = bench.find_profile("issue", key, "0" * 16)
    assert hit is None and stale["same_schedule"] and stale["profile_meta"]["source_hash"] == m["source_hash"]
    assert stale["this_run"]["source_hash"] == "0" * 16
    # another schedule / viewport of the same workload: not used either
    for change in ({"viewport": 2048}, {"frames_per_launch": 7}, {"round_budget": 10, "frames_in_flight": 5}):
        hit, stale = bench.find_profile("issue", dict(key, **change), m["source_hash"])
        assert hit is None and stale is not None and stale["this_run"]["workload"] == "shells2048"
    # a workload nobody profiled: nothing at all
    hit, stale = bench.find_profile("issue", dict(key, workload="no_such_workload"), m["source_hash"])
    assert hit is None and stale is None


def test_committed_profiles_belong_to_the_committed_kernel_sources():
    """The latest profiles/r<N> is regenerated as the last act of a round, so its hash is the hash of csrc/ as committed.  While
    kernels are being worked on the two differ and bench.py says so in its line ("stale_profile"); here that is a
    warning in the test summary, not a failure."""
    import warnings
    h = bench.source_hash()
    for kind in ("issue", "traffic"):
        for path, e in _committed(kind):
            if e["meta"]["source_hash"] != h:
                warnings.warn("%s was measured from kernel sources %s, the tree holds %s: regenerate profiles/%s"
                              % (os.path.basename(path), e["meta"]["source_hash"], h, _latest_round()))
