"""The evidence the bench line and the documents point at is in the tree."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_traffic_json_sources_exist():
    d = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    keys = [k for k in d if not k.startswith("_")]
    assert {"2", "2_2048blocks", "3", "4", "5"} <= set(keys)
    for k in keys:
        assert (ROOT / d[k]["source"]).is_file(), d[k]["source"]
        assert d[k]["bytes"] > 0 and d[k]["kernels"]


def test_documents_cite_existing_profiles():
    missing = []
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md", "profiles/r02_summary.md", "profiles/r03_summary.md",
                "profiles/r04_summary.md"):
        if not (ROOT / doc).exists():
            continue
        text = (ROOT / doc).read_text()
        for name in set(re.findall(r"profiles/(r0[1-4]_[A-Za-z0-9_]+\.(?:txt|json|csv|md))", text)):
            if not (ROOT / "profiles" / name).exists():
                missing.append((doc, name))
    assert not missing, missing
