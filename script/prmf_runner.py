#!/usr/bin/env python
"""Drop-in for the reference's script/prmf_runner.py: same flags and output files,
solve on the MI355X (see prmf_amd/runner.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prmf_amd.runner import main  # noqa: E402

if __name__ == "__main__":
    main()
