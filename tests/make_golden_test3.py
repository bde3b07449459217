"""Build tests/golden/test3_transcript.json.gz from the reference's golden transcript
tests/integration/test3.t (run in the build container, where /root/reference exists).

test3.sh:  seed 10; load -i test3_families.txt -p 0.01; tree ...; lambda -s -t (((2,2)1,(1,1)1)1,1);
           lambda -l 0.0017; rootdist -i fly.table; genfamily rndtree/rnd -t 10;
           lhtest -d rndtree -l 0.0017 -t (((2,2)1,(1,1)1)1,1) -o lh2.out
The transcript is turned into a flat list of events (numbers only):
   ["families", n] ["root_range", lo, hi] ["family_range", lo, hi] ["poisson", lambda, score, iters]
   ["eval", [lambda...], score] ["result", iters, [lambda...], score]
test3_families.txt is byte-identical to example/example_data.tab (already a fixture); fly.table is copied
as data.  The same parser (tests/_transcript.py) reads this repo's log, so the two event lists compare 1:1.
"""
import gzip
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from tests._transcript import parse_events  # noqa: E402

REF = "/root/reference/tests/integration"


def main():
    a = open(os.path.join(REF, "test3_families.txt"), "rb").read()
    b = open(os.path.join(HERE, "golden", "example_data.tab"), "rb").read()
    assert a == b, "test3_families.txt is expected to equal example_data.tab"
    shutil.copyfile(os.path.join(REF, "fly.table"), os.path.join(HERE, "golden", "fly.table"))
    text = open(os.path.join(REF, "test3.t")).read()
    # cram transcripts indent every output line by two spaces
    lines = [l[2:] if l.startswith("  ") else l for l in text.splitlines()]
    ev = parse_events("\n".join(lines))
    out = {"source": "tests/integration/test3.t (CAFE v4.1 transcript), tests/integration/test3.sh", "events": ev}
    with gzip.open(os.path.join(HERE, "golden", "test3_transcript.json.gz"), "wt") as f:
        json.dump(out, f)
    kinds = {}
    for e in ev:
        kinds[e[0]] = kinds.get(e[0], 0) + 1
    print(len(ev), "events", kinds)


if __name__ == "__main__":
    main()
