#!/usr/bin/env python3
"""Fuzzers of the engine against the oracle (GPU box; test infrastructure, not part of the test suite except where
tests/test_gpu_round3.py runs a few hundred cases of three of them).  One entry point, one sub-command per suite:

  python3 tools/soak.py frames    [frames per configuration]   random frames of seven size / threshold / octave configurations, batch path
  python3 tools/soak.py describe                               descriptor-only calls (all scales, borders, angles, both patterns, flags), dense detection, 4K
  python3 tools/soak.py ordered                                ordered path: thresholds 1..19, multi-layer no-scale-NMS, ComputeScale
  python3 tools/soak.py callspace [cases] [seed]               image sides 9..1100, thresholds 1..140, 0..6 octaves, five content kinds, host calls and batches
  python3 tools/soak.py options   [cases] [seed]               masks, suppressScaleNonmaxima=false, post-filters, invariance flags, pattern versions / scales
  python3 tools/soak.py matcher   [cases] [seed]               Hamming matcher: set sizes, 1..6 train images, descriptor lengths 16..224, masks, k, radii
  python3 tools/soak.py large     [cases] [seed]               16-bit image functions, frames of 2000..4600 px, ComputeScale lists
  python3 tools/soak.py threads   [iterations] [watchdog s]    eight host threads with their own contexts and kinds of work at once (one-frame 4K, 64-frame
                                                               batch, dense, odd sizes, host-to-host), every iteration against the oracle; a hang kills the child
  python3 tools/soak.py hostpaths [cases] [seed]               round 6's host-side paths: multi-image calls, host-to-host batches with exact / short row capacities, the pool from six threads
  python3 tools/soak.py all                                    every suite with its defaults

Every case is compared bit-exactly with the oracle (or both sides must agree that the reference has no defined result).
The suites live in tools/soak_cases/<name>.py (importable: tools/repro_*.py rebuild single cases from them)."""
import os
import subprocess
import sys

SUITES = ["frames", "describe", "ordered", "callspace", "options", "matcher", "large", "threads", "hostpaths"]
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak_cases")


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in SUITES + ["all"]:
        print(__doc__)
        return 2
    names = SUITES if sys.argv[1] == "all" else [sys.argv[1]]
    rest = sys.argv[2:]
    for name in names:  # (a process of its own per suite: they fork their oracle workers before HIP is initialised)
        rc = subprocess.call([sys.executable, os.path.join(HERE, name + ".py")] + (rest if len(names) == 1 else []))
        if rc:
            return rc
    return 0


if __name__ == "__main__":
    sys.exit(main())
