"""The product package never imports, links or executes anything under oracle/ (or the reference)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_sources_do_not_touch_oracle_or_reference():
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|oracle[/.]|/root/reference", re.M)
    checked = 0
    for base in ("partmanip_amd", "algorithms"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dp, f)).read()
                    assert not bad.search(src), f"{dp}/{f} references the oracle/reference"
                    checked += 1
    assert checked > 10


def test_oracle_is_only_used_by_tests_smoke_and_bench():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in ("bench.py", "__graft_entry__.py"):
            assert not pat.search(open(os.path.join(ROOT, f)).read()), f
