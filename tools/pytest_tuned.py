#!/usr/bin/env python
"""Run pytest with kernel-variant knobs set first: tools/pytest_tuned.py key=value [key=value ...] -- <pytest args>"""
import sys

import pytest

sys.path.insert(0, ".")
from simpledet_amd._lib import lib  # noqa: E402

i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    lib().set_tuning(k, int(v))
sys.exit(pytest.main(sys.argv[i + 1:]))
