"""Parser shared by the golden-transcript builder and the tests: CAFE log text -> list of numeric events."""
import re

_NUM = r"[-+]?(?:inf|nan|\d+\.?\d*(?:[eE][-+]?\d+)?)"
_PATTERNS = [
    ("families", re.compile(r"^The number of families is (\d+)")),
    ("root_range", re.compile(r"^Root Family size : (\d+) ~ (\d+)")),
    ("family_range", re.compile(r"^Family size : (\d+) ~ (\d+)")),
    ("poisson_iters", re.compile(r"^Empirical Prior Estimation Result: \((\d+) iterations\)")),
    ("poisson", re.compile(r"^Poisson lambda: (%s) & Score: (%s)" % (_NUM, _NUM))),
    ("result_iters", re.compile(r"^Lambda Search Result: (\d+)")),
    ("eval", re.compile(r"^\.?Lambda : ([-+0-9.,einfa]+) & Score: (%s)" % _NUM)),
]


def parse_events(text):
    ev = []
    pending_iters = None
    pending_result = None
    for line in text.splitlines():
        for kind, rx in _PATTERNS:
            m = rx.match(line)
            if not m:
                continue
            if kind == "families":
                ev.append(["families", int(m.group(1))])
            elif kind in ("root_range", "family_range"):
                ev.append([kind, int(m.group(1)), int(m.group(2))])
            elif kind == "poisson_iters":
                pending_iters = int(m.group(1))
            elif kind == "poisson":
                ev.append(["poisson", float(m.group(1)), float(m.group(2)), pending_iters])
                pending_iters = None
            elif kind == "result_iters":
                pending_result = int(m.group(1))
            elif kind == "eval":
                lam = [float(x) for x in m.group(1).split(",")]
                if pending_result is not None:
                    ev.append(["result", pending_result, lam, float(m.group(2))])
                    pending_result = None
                else:
                    ev.append(["eval", lam, float(m.group(2))])
            break
    return ev
