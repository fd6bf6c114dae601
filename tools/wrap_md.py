#!/usr/bin/env python3
"""Re-flow a markdown file to <= WIDTH columns: paragraphs and list items are wrapped (continuation lines indented under the item text), tables whose rows
exceed WIDTH are turned into lists (first column = item head, the other columns as `header: cell` continuation paragraphs), code fences and short lines are
left alone.  Usage: python tools/wrap_md.py FILE [WIDTH]  (rewrites FILE in place)."""
import re
import sys
import textwrap

WIDTH = int(sys.argv[2]) if len(sys.argv) > 2 else 160


def wrap(text, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells(row):
    # split on unescaped pipes
    parts = re.split(r"(?<!\\)\|", row.strip())
    return [p.strip() for p in parts[1:-1]]


def main(path):
    src = open(path).read().split("\n")
    out, i, fence = [], 0, False
    while i < len(src):
        ln = src[i]
        if ln.lstrip().startswith("```"):
            fence = not fence
            out.append(ln)
            i += 1
            continue
        if fence or len(ln) <= WIDTH and not ln.lstrip().startswith("|"):
            out.append(ln)
            i += 1
            continue
        if ln.lstrip().startswith("|"):
            j = i
            while j < len(src) and src[j].lstrip().startswith("|"):
                j += 1
            block = src[i:j]
            if max(len(b) for b in block) <= WIDTH:
                out += block
            else:
                head = cells(block[0])
                for row in block[2:]:
                    c = cells(row)
                    if not c:
                        continue
                    out.append(wrap(f"**{c[0]}**" if not c[0].startswith("**") else c[0], "- ", "  "))
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            out.append(wrap(f"*{h}*: {v}" if h else v, "  ", "  "))
                out.append("")
            i = j
            continue
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", ln)
        if m:
            ind = " " * len(m.group(0))
            out.append(wrap(ln[len(m.group(0)):], m.group(0), ind))
        else:
            lead = re.match(r"^\s*", ln).group(0)
            out.append(wrap(ln, lead, lead))
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
