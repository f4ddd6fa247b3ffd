"""Hash of the kernel sources that survives comments and layout.

profiles/pmc*.json are stamped with it (tools/make_traffic.py) and bench.py reports the PMC-derived
figures as stale when it differs.  Rounds 1-5 hashed the text, so a note added to a source file cost
five PMC legs of lease time; this one hashes the token stream: comments removed (// and /* */, string
and character literals respected -- the inline asm lives in strings), whitespace runs collapsed."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _segments(text):
    """(is_literal, text) pieces of a C++ source with its comments replaced by a blank."""
    i, n, code = 0, len(text), []
    while i < n:
        c = text[i]
        if c == "/" and i + 1 < n and text[i + 1] == "/":
            j = text.find("\n", i)
            # (a line comment that ends in a backslash continues on the next line)
            while j != -1 and text[j - 1] == "\\":
                j = text.find("\n", j + 1)
            i = n if j == -1 else j
            code.append(" ")
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j == -1 else j + 2
            code.append(" ")
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            yield False, "".join(code)
            code = []
            yield True, text[i:j + 1]
            i = j + 1
        else:
            code.append(c)
            i += 1
    yield False, "".join(code)


def normalised(text):
    """Comments removed, whitespace runs outside string / character literals collapsed."""
    out = []
    for lit, seg in _segments(text):
        if lit:
            out.append(seg)
        else:
            lead = " " if seg[:1].isspace() else ""
            tail = " " if seg[-1:].isspace() and len(seg.strip()) else ""
            out.append(lead + " ".join(seg.split()) + tail)
    return "".join(out).strip()


def source_sha():
    h = hashlib.sha256()
    src = os.path.join(ROOT, "peakachu_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(normalised(open(os.path.join(src, name), encoding="utf-8").read()).encode("utf-8"))
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha())
