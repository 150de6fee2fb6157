"""pytest over the test ids listed in a file (ids with spaces / brackets survive): python tools/run_ids.py ids.txt [pytest args]"""
import sys

import pytest

ids = [l.strip() for l in open(sys.argv[1]) if l.strip()]
sys.exit(pytest.main(ids + sys.argv[2:]))
