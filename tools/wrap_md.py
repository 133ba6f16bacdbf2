#!/usr/bin/env python3
"""tools/wrap_md.py FILE WIDTH -- re-wrap the paragraphs and bullets of a Markdown file to WIDTH BYTES per line (tables, code blocks and
headings are left alone; a bullet's continuation lines are indented under its text)."""
import re
import sys


def blen(s):
    return len(s.encode("utf-8"))


def wrap(words, width, first, rest):
    lines, cur = [], first
    empty = True
    for w in words:
        if not empty and blen(cur) + 1 + blen(w) > width:
            lines.append(cur)
            cur, empty = rest, True
        cur = cur + w if empty else cur + " " + w
        empty = False
    lines.append(cur)
    return lines


def main(path, width):
    src = open(path, encoding="utf-8").read().split("\n")
    out, i, in_code = [], 0, False
    while i < len(src):
        line = src[i]
        if line.lstrip().startswith("```"):
            in_code = not in_code
            out.append(line)
            i += 1
            continue
        if in_code or not line.strip() or line.lstrip().startswith(("#", "|")) or re.match(r"^\s*(---|===)", line):
            out.append(line)
            i += 1
            continue
        m = re.match(r"^(\s*)([*\-] |\d+\. )?", line)
        indent, bullet = m.group(1), m.group(2) or ""
        first = indent + bullet
        rest = indent + " " * len(bullet)
        block = [line[len(first):]]
        i += 1
        while i < len(src):
            nxt = src[i]
            if (not nxt.strip() or nxt.lstrip().startswith(("#", "|", "```")) or re.match(r"^\s*([*\-] |\d+\. )", nxt)):
                break
            block.append(nxt.strip())
            i += 1
        out.extend(wrap(" ".join(block).split(), width, first, rest))
    open(path, "w", encoding="utf-8").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
