#!/usr/bin/env python3
"""wrap_comments.py -- keep the C++ / HIP sources inside `width` columns where that is a matter of COMMENTS: a trailing `// ...` that pushes its line past the width
moves onto lines of its own above the code it belongs to, and a comment-only line that is more than a few columns too long is re-flowed.  Code is never touched (a code line that is too long by
itself stays), nor are preprocessor lines, macro continuations, or comments that look like tables / drawings (runs of spaces inside the text).
Usage: python tools/wrap_comments.py [--width 160] [--check] files...      (--check: list the lines that would change, write nothing)"""
import argparse
import re
import sys
import textwrap


def comment_start(line):
    """index of the `//` that starts a trailing comment, outside string and character literals; -1 if none"""
    i, n, q = 0, len(line), None
    while i < n:
        c = line[i]
        if q:
            if c == "\\":
                i += 2
                continue
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c == "/" and i + 1 < n and line[i + 1] == "/":
            return i
        elif c == "/" and i + 1 < n and line[i + 1] == "*":
            return -1                                                   # a block comment on the line: leave it alone
        i += 1
    return -1


def looks_drawn(text):
    return re.search(r"\S {3,}\S", text) is not None                      # columns kept apart by runs of spaces: a table or a drawing


def balanced(body, w):
    """wrapped to at most w columns, the lines about equally long (no two-word last line)"""
    parts = textwrap.wrap(body, w, break_long_words=False, break_on_hyphens=False)
    n = len(parts)
    if n > 1:
        for ww in range(max(40, -(-len(body) // n)), w + 1):
            q = textwrap.wrap(body, ww, break_long_words=False, break_on_hyphens=False)
            if len(q) <= n:
                return q
    return parts


def flow(indent, text, width):
    """`text` (without its //) as comment lines of at most `width` columns"""
    lead = re.match(r"\s*", text).group(0)
    body = text.strip()
    if not body:
        return [indent + "//"]
    prefix = indent + "//" + (lead if lead else " ")
    w = max(40, width - len(prefix))
    parts = balanced(body, w)
    return [prefix + p for p in parts]


def process(lines, width, slack=20, a_block=False):
    out, changed = [], []
    prev_cont = False
    skip = 0
    in_block = False
    for ln, raw in enumerate(lines, 1):
        if skip:
            skip -= 1
            continue
        line = raw.rstrip("\n")
        in_block_before = in_block
        if "/*" in line and "*/" not in line[line.index("/*"):]:
            in_block = True
        elif in_block and "*/" in line:
            in_block = False
        cont = line.rstrip().endswith("\\")
        if len(line) <= width or prev_cont or cont or line.lstrip().startswith("#"):
            out.append(line); prev_cont = cont
            continue
        prev_cont = cont
        stripped = line.lstrip()
        indent = line[:len(line) - len(stripped)]
        if a_block and in_block_before and stripped.startswith("* ") and "*/" not in stripped:   # a line of a /* ... */ block written with a star in front of every line
            text = stripped[2:]
            if looks_drawn(text):
                out.append(line)
                continue
            lead = re.match(r"\s*", text).group(0)
            w = max(40, width - len(indent) - 2 - len(lead))
            parts = balanced(text.strip(), w)
            out.extend(indent + "* " + lead + q for q in parts); changed.append(ln)
            continue
        if stripped.startswith("//"):
            text = stripped[2:]
            if stripped.startswith("///") or looks_drawn(text) or len(line) <= width + slack:    # (a line of a paragraph that is a few columns over stays: splitting it would leave a ragged paragraph)
                out.append(line)
                continue
            new = flow(indent, text, width)
            out.extend(new); changed.append(ln)
            continue
        pos = comment_start(line)
        if pos <= 0:
            # a declaration with a closed /* ... */ behind it: the comment moves above, as a block of its own
            m = re.match(r"^(\s*)(\S.*?;)\s*/\*\s*(.*?)\s*\*/\s*$", line)
            if m and "/*" not in m.group(2) and not in_block_before and not looks_drawn(m.group(3)):
                parts = balanced(m.group(3), max(40, width - len(m.group(1)) - 6))
                if len(parts) == 1:
                    out.append(m.group(1) + "/* " + parts[0] + " */")
                else:
                    out.append(m.group(1) + "/* " + parts[0])
                    out.extend(m.group(1) + " * " + q for q in parts[1:-1])
                    out.append(m.group(1) + " * " + parts[-1] + " */")
                out.append(m.group(1) + m.group(2)); changed.append(ln)
                continue
            out.append(line)
            continue
        code, text = line[:pos].rstrip(), line[pos + 2:]
        if not code.strip() or looks_drawn(text) and len(code) > width - 40:
            out.append(line)
            continue
        # a trailing comment that goes on over comment-only lines indented well past the code: one comment, moved as a whole
        k = ln
        while k < len(lines):
            nxt = lines[k].rstrip("\n"); ns = nxt.lstrip()
            if ns.startswith("//") and not ns.startswith("///") and len(nxt) - len(ns) >= len(indent) + 8 and not looks_drawn(ns[2:]):
                text = text.rstrip() + " " + ns[2:].strip(); k += 1
            else:
                break
        skip = k - ln
        out.extend(flow(indent, text, width)); out.append(code); changed.append(ln)
    return out, changed


def break_points(code):
    """(position after a top-level-ish separator, kind) outside string / character literals: ', ' inside parentheses, '; ' between statements"""
    pts, q, depth, i, n = [], None, 0, 0, len(code)
    while i < n:
        c = code[i]
        if q:
            if c == "\\":
                i += 2
                continue
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        elif c == "," and depth >= 1 and i + 1 < n and code[i + 1] == " ":
            pts.append((i + 2, ","))
        elif c == ";" and depth == 0 and i + 1 < n and code[i + 1] == " ":
            pts.append((i + 2, ";"))
        i += 1
    return pts


def wrap_code(line, width):
    """a code line broken behind commas of argument lists / between statements; None if it cannot be brought under the width sensibly"""
    stripped = line.lstrip()
    indent = line[:len(line) - len(stripped)]
    pos = comment_start(line)
    if pos >= 0 or "/*" in line:
        return None                                                     # (comments are the first pass's business)
    out, cur, first = [], line, True
    while len(cur) > width:
        body = cur
        cands = [(p, k) for p, k in break_points(body) if len(indent) + 16 < p <= width]
        if not cands:
            return None
        p, k = cands[-1]
        out.append(body[:p].rstrip())
        cur = indent + ("        " if k == "," else "    ") + body[p:].lstrip()
        first = False
        if len(out) > 6:
            return None
    out.append(cur)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--code", type=int, default=0, help="also break CODE lines longer than this many columns behind commas of argument lists / between statements (0: leave code alone)")
    ap.add_argument("files", nargs="+")
    a = ap.parse_args()
    total = 0
    for f in a.files:
        with open(f) as fh:
            lines = fh.readlines()
        out, changed = process(lines, a.width)
        if a.code:
            out2, prev_cont = [], False
            for ln, line in enumerate(out, 1):
                cont = line.rstrip().endswith("\\")
                w = None if (len(line) <= a.code or prev_cont or cont or line.lstrip().startswith("#")) else wrap_code(line, a.width)
                prev_cont = cont
                if w:
                    out2.extend(w); changed.append(ln)
                else:
                    out2.append(line)
            out = out2
        total += len(changed)
        if a.check:
            print("%-44s %4d lines would change, %4d stay longer than %d" % (f, len(changed), sum(len(x) > a.width for x in out), a.width))
        elif changed:
            with open(f, "w") as fh:
                fh.write("\n".join(out) + "\n")
            print("%-44s %4d lines re-flowed, %4d stay longer than %d" % (f, len(changed), sum(len(x) > a.width for x in out), a.width))
    return 0


if __name__ == "__main__":
    sys.exit(main())
