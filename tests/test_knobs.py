"""The library's configuration surface is ONE file (csrc/knobs.hpp): every environment variable it honours in a shipped build is in the
`pub` list there, described in that header's comment, and mentioned in the docs a user reads; nothing else in csrc/ calls getenv."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "noble-curves_amd", "csrc")


def _read(*p):
    with open(os.path.join(*p)) as f:
        return f.read()


def test_public_knobs_are_listed_described_and_documented():
    src = _read(CSRC, "knobs.hpp")
    m = re.search(r"pub\[\]\s*=\s*\{([^}]*)\}", src)
    assert m, "the list of public knobs moved"
    pub = re.findall(r'"(NCG_[A-Z0-9_]+)"', m.group(1))
    assert len(pub) == len(set(pub)) and "NCG_LANE_QUEUES" in pub and "NCG_NO_FINISH_THREADS" in pub
    header = src.split("#pragma once")[0]
    always = header.split("A/B builds only")[0]
    docs = _read(ROOT, "INTEGRATION.md") + _read(ROOT, "DESIGN.md")
    for k in pub:
        assert k in always, "%s is honoured by the shipped library but not described under 'always honoured'" % k
    for k in ("NCG_LANE_QUEUES", "NCG_NO_FINISH_THREADS"):     # the two that change what the library does to the embedding process
        assert k in _read(ROOT, "INTEGRATION.md"), k
    for k in pub:
        assert k in docs or k in ("NCG_TIMING",), "%s is not mentioned in INTEGRATION.md / DESIGN.md" % k


def test_only_knobs_hpp_reads_the_environment():
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".hpp")) or name == "knobs.hpp":
            continue
        text = _read(CSRC, name)
        assert "getenv" not in text, "%s reads the environment directly (use ncg::knob)" % name
