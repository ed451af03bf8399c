"""The documents quote measurements by file: every `profiles/...`, `tools/...` and `tests/...` path they name has to exist."""
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_paths_named_in_the_documents_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", "tools/probe/README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`((?:profiles|tools|tests|oracle|include)/[A-Za-z0-9_./-]+)`", text):
            path = m.group(1).rstrip(".")
            if "…" in path or "*" in path or path.endswith("/_ref") or path.endswith(".so"):       # (abbreviated, globbed, built)
                continue
            path = path.split("::")[0]
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing
