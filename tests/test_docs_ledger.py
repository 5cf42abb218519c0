"""CPU: docs/PARITY.md is the document a maintainer trusts -- it may not cite what the tree no longer holds.

Every test name, file path and numeric tolerance the ledger quotes must exist under tests/ (or on disk); statements of earlier
rounds that the tests have since replaced by plain equality (`N mod 64`, `atol 2e-7`, "exact vs the DOUBLED batch" as the bar of
forward / optimize) fail here until the ledger is brought up to date."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEDGER = os.path.join(ROOT, "docs", "PARITY.md")


def _tests_text():
    return "\n".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "*.py")))
                     if os.path.basename(f) != os.path.basename(__file__))


def test_every_test_and_file_the_ledger_names_exists():
    text, tests = open(LEDGER).read(), _tests_text()
    names = set(re.findall(r"`((?:tests/)?(?:test_|stress_)[A-Za-z0-9_\.\*]+)`", text))
    assert names, "the ledger names no tests?"
    for n in sorted(names):
        n = n[len("tests/"):] if n.startswith("tests/") else n
        if n.endswith(".py"):
            assert glob.glob(os.path.join(ROOT, "tests", n)), "ledger cites tests/%s: no such file" % n
        else:
            stem = n.rstrip("*")
            found = re.search(r"def %s" % re.escape(stem), tests) or (n.endswith("*") and glob.glob(os.path.join(ROOT, "tests", n + ".py")))
            assert found, "ledger cites %s: no test (or test file) of that name under tests/" % n
    for p in sorted(set(re.findall(r"`((?:profiles|tools|tests|docs|oracle)/[A-Za-z0-9_\./-]+)`", text))):
        assert os.path.exists(os.path.join(ROOT, p)), "ledger cites %s: not in the tree" % p
    helpers = set(re.findall(r"`(?:conftest\.)?(check_[a-z_]+)`", text))
    for h in sorted(helpers):
        assert re.search(r"def %s" % h, tests), "ledger cites helper %s: not under tests/" % h


def test_every_tolerance_the_ledger_quotes_is_one_a_test_still_applies():
    text, tests = open(LEDGER).read(), _tests_text()
    # numeric tolerances: "atol 2e-7", "atol=1e-6", "≤ 2e-7", "<= 1.2e-7" ...
    quoted = set(re.findall(r"atol[ =]*([0-9.]+e-?[0-9]+)", text)) | set(re.findall(r"[≤<]=? ?([0-9.]+e-[0-9]+)", text))
    applied = set(re.findall(r"atol=([0-9.]+e-?[0-9]+)", tests)) | set(re.findall(r"[<≤]=? ?([0-9.]+e-[0-9]+)", tests))
    norm = lambda s: "%.6g" % float(s)          # noqa: E731
    applied = {norm(a) for a in applied}
    for q in sorted(quoted):
        assert norm(q) in applied, "ledger quotes a tolerance of %s that no test under tests/ applies" % q
    # the block of ATen's outer sum: whatever `mod K` the ledger names for trailing columns / points must be the K of the tests
    for k in sorted(set(re.findall(r"mod (\d+)`? (?:points|columns)", text))):
        assert re.search(r"mod %s\b" % k, tests), "ledger says 'mod %s': the tests say something else" % k
    # bars of earlier rounds that plain equality has replaced
    for stale in ("N mod 64", "vs the DOUBLED batch", "exact vs the doubled batch", "exact vs the recomposed batch"):
        assert stale not in text, "ledger still carries the round-4 statement %r" % stale
