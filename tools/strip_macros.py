#!/usr/bin/env python
"""Resolve `#ifdef X / #ifndef X / #else / #endif` blocks for macros known to be undefined (one-off source clean-up)."""
import re
import sys


def strip(text, undefined):
    out, stack = [], []  # stack entries: (kind, keep_now, is_target)
    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(\w*)", line)
        emit = all(k for _, k, _ in stack)
        if m:
            d, name = m.group(1), m.group(2)
            if d in ("ifdef", "ifndef") and name in undefined:
                stack.append((d, d == "ifndef", True))
                continue
            if d in ("ifdef", "ifndef", "if"):
                stack.append((d, True, False))
                if emit:
                    out.append(line)
                continue
            if d in ("else", "elif"):
                kind, keep, tgt = stack[-1]
                if tgt:
                    stack[-1] = (kind, not keep, True)
                    continue
                if all(k for _, k, _ in stack[:-1]):
                    out.append(line)
                continue
            if d == "endif":
                kind, keep, tgt = stack.pop()
                if tgt:
                    continue
                if all(k for _, k, _ in stack):
                    out.append(line)
                continue
        if emit:
            out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    path, names = sys.argv[1], set(sys.argv[2:])
    src = open(path).read()
    open(path, "w").write(strip(src, names))
