#!/usr/bin/env python3
"""Re-wrap the prose of a markdown file at 118 columns (tables, code blocks, headings and list structure are kept).
    python tools/wrap_md.py FILE..."""
import re
import sys
import textwrap


def wrap(text, width=118):
    out, para, in_code = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([*\-+]|\d+\.)\s+", first)
        if m:
            indent = " " * len(m.group(0))
            body = " ".join([first[len(m.group(0)):].strip()] + [p.strip() for p in para[1:]])
            out.extend(textwrap.wrap(body, width, initial_indent=m.group(0), subsequent_indent=indent,
                                     break_long_words=False, break_on_hyphens=False))
        else:
            lead = re.match(r"^\s*", first).group(0)
            body = " ".join(p.strip() for p in para)
            out.extend(textwrap.wrap(body, width, initial_indent=lead, subsequent_indent=lead, break_long_words=False,
                                     break_on_hyphens=False))
        para.clear()

    for line in text.split("\n"):
        if line.strip().startswith("```"):
            flush()
            in_code = not in_code
            out.append(line)
            continue
        if in_code or line.startswith("|") or line.startswith("#") or not line.strip() or line.startswith(">"):
            flush()
            out.append(line)
            continue
        if re.match(r"^\s*([*\-+]|\d+\.)\s+", line):      # a new list item starts a new paragraph
            flush()
            para.append(line)
            continue
        if para and re.match(r"^\s+", line) is None and re.match(r"^\s*([*\-+]|\d+\.)\s+", para[0]):
            flush()                                          # unindented text after a list item: a new paragraph
        para.append(line)
    flush()
    return "\n".join(out)


for path in sys.argv[1:]:
    src = open(path).read()
    open(path, "w").write(wrap(src))
    print(path, "lines over 120 outside tables:",
          sum(1 for l in wrap(src).split("\n") if len(l) > 120 and not l.startswith("|")))
