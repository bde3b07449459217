"""Regenerate / verify the test1 and test2 entries of tests/golden/transcripts.json from the reference's golden
transcripts tests/integration/test1.t and test2.t (run in the build container, where /root/reference exists), with
the same parser the tests use on this repo's log (tests/_transcript.py).

    python tests/make_golden_transcripts.py          # verify the committed numbers
    python tests/make_golden_transcripts.py --write  # rewrite lambda_score / search_result / poisson_* in place
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from tests._transcript import parse_events  # noqa: E402

REF = "/root/reference/tests/integration"


def extract(name):
    text = open(os.path.join(REF, name + ".t")).read()
    lines = [l[2:] if l.startswith("  ") else l for l in text.splitlines()]
    ev = parse_events("\n".join(lines))
    out = {}
    po = next(e for e in ev if e[0] == "poisson")
    out["poisson_lambda"], out["poisson_score"], out["poisson_iters"] = po[1], po[2], po[3]
    out["n_families"] = next(e for e in ev if e[0] == "families")[1]
    out["root_range"] = next(e for e in ev if e[0] == "root_range")[1:]
    out["family_range"] = next(e for e in ev if e[0] == "family_range")[1:]
    out["lambda_score"] = [[e[1][0], e[2]] for e in ev if e[0] == "eval"]
    res = next(e for e in ev if e[0] == "result")
    out["search_result"] = {"lambda": res[2][0], "score": res[3], "iters": res[1]}
    return out


def main():
    path = os.path.join(HERE, "golden", "transcripts.json")
    t = json.load(open(path))
    ok = True
    for name in ("test1", "test2"):
        got = extract(name)
        for k, v in got.items():
            cur = t[name].get(k)
            same = json.dumps(cur) == json.dumps(v) or (k == "lambda_score" and cur == v[:len(cur)])
            if not same:
                ok = False
                print(name, k, "differs:", str(cur)[:80], "vs", str(v)[:80])
            if "--write" in sys.argv:
                t[name][k] = v
    if "--write" in sys.argv:
        json.dump(t, open(path, "w"), indent=1)
        print("written")
    print("transcripts.json matches the reference transcripts" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
