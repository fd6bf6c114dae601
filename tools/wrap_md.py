#!/usr/bin/env python3
"""Re-flow a markdown file to <= WIDTH columns.  Paragraphs and list items are joined and wrapped again (continuation lines indented under the item text),
tables whose rows exceed WIDTH are turned into lists (first column = item head, the other columns as `header: cell` continuation paragraphs), headings, code
fences, tables that fit and html are left alone.  Idempotent.  Usage: python tools/wrap_md.py FILE [WIDTH]  (rewrites FILE in place)."""
import re
import sys
import textwrap

WIDTH = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ITEM = re.compile(r"^(\s*)([-*+]|\d+\.)\s+")


def wrap(text, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells(row):
    parts = re.split(r"(?<!\\)\|", row.strip())
    return [p.strip() for p in parts[1:-1]]


def special(ln):
    s = ln.lstrip()
    return (not s) or s.startswith(("#", "|", "```", "<", ">")) or bool(ITEM.match(ln)) or set(s) <= set("-=*_ ")


def main(path):
    src = open(path).read().split("\n")
    out, i, fence = [], 0, False
    while i < len(src):
        ln = src[i]
        s = ln.lstrip()
        if s.startswith("```"):
            fence = not fence
            out.append(ln)
            i += 1
            continue
        if fence or not s or s.startswith(("#", "<", ">")) or set(s) <= set("-=*_ "):
            out.append(ln)
            i += 1
            continue
        if s.startswith("|"):
            j = i
            while j < len(src) and src[j].lstrip().startswith("|"):
                j += 1
            block = src[i:j]
            if max(len(b) for b in block) <= WIDTH + 10:
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
                            out.append(wrap(f"*{h}*: {v}" if h else v, "  - ", "    "))
                out.append("")
            i = j
            continue
        # a paragraph or a list item: gather its continuation lines (not special, indented at least as much as the item text / the paragraph)
        m = ITEM.match(ln)
        first = m.group(0) if m else re.match(r"^\s*", ln).group(0)
        rest = " " * len(first)
        text = [ln[len(first):]]
        j = i + 1
        while j < len(src) and not special(src[j]) and (len(src[j]) - len(src[j].lstrip())) >= (len(rest) if m else len(first)):
            text.append(src[j].strip())
            j += 1
        out.append(wrap(" ".join(text), first, rest))
        i = j
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
