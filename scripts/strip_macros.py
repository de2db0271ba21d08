"""Removes the regions of named preprocessor switches from source files, keeping the branch a build WITHOUT the switch compiles
(round 6: the timing / ablation / A-B bodies left the product translation units this way; the removed text lives on as patch files
under scripts/variants/, applied to a scratch copy of the sources by scripts/build_variant.sh).
    python scripts/strip_macros.py --undef A,B --value C=0 [--drop-calls X_MARK,Y_MARK] file...
Understands #ifdef X / #ifndef X / #if X == n / #if defined(X) && X == n / #else / #endif for the NAMED switches only; every other
conditional is left alone. --drop-calls removes the definitions `#define NAME(...) ...` and the statements `NAME(...);` of marker macros."""
import argparse, re, sys

ap = argparse.ArgumentParser()
ap.add_argument("--undef", default="")
ap.add_argument("--value", action="append", default=[])
ap.add_argument("--drop-calls", default="")
ap.add_argument("files", nargs="+")
a = ap.parse_args()
undef = set(x for x in a.undef.split(",") if x)
values = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.value)
drop = [x for x in a.drop_calls.split(",") if x]
names = undef | set(values)


def evaluate(line):
    """None: not ours. Otherwise the truth of the condition in a build without the switches."""
    s = line.strip()
    m = re.match(r"#\s*ifdef\s+(\w+)", s)
    if m:
        return (m.group(1) in values) if m.group(1) in names else None
    m = re.match(r"#\s*ifndef\s+(\w+)", s)
    if m:
        return (m.group(1) not in values) if m.group(1) in names else None
    m = re.match(r"#\s*if\s+(?:defined\((\w+)\)\s*&&\s*)?(\w+)\s*==\s*(\d+)", s)
    if m and m.group(2) in names:
        if m.group(2) in undef:
            return 0 == int(m.group(3)) if m.group(1) is None else False
        return values[m.group(2)] == int(m.group(3))
    return None


for path in a.files:
    out, stack = [], []   # stack entries: [ours, keep_now, parent_keep]
    for line in open(path).read().split("\n"):
        s = line.strip()
        keep = all(e[1] for e in stack)
        if re.match(r"#\s*if", s):
            v = evaluate(line)
            if v is None:
                stack.append([False, True, keep])
                if keep:
                    out.append(line)
            else:
                stack.append([True, bool(v), keep])
            continue
        if re.match(r"#\s*else\b", s) and stack:
            if stack[-1][0]:
                stack[-1][1] = not stack[-1][1]
            elif all(e[1] for e in stack):
                out.append(line)
            continue
        if re.match(r"#\s*endif\b", s) and stack:
            e = stack.pop()
            if not e[0] and all(x[1] for x in stack):
                out.append(line)
            continue
        if keep:
            out.append(line)
    assert not stack, path
    text = "\n".join(out)
    if drop:   # `if (cond) MARK(i);` and `if (cond) MARK(i); else MARK(j);` on a line of their own go as a whole
        alt = "|".join(drop)
        text = re.sub(r"^[ \t]*if \([^\n]*\) (?:%s)\(\w+\);(?: else (?:%s)\(\w+\);)?[ \t]*\n" % (alt, alt), "", text, flags=re.M)
    for n in drop:
        text = re.sub(r"^[ \t]*#[ \t]*define[ \t]+%s\b[^\n]*\n" % n, "", text, flags=re.M)
        text = re.sub(r"^[ \t]*%s\([^;\n]*\);[ \t]*(//[^\n]*)?\n" % n, "", text, flags=re.M)    # a statement on a line of its own
        text = re.sub(r"[ \t]*(?<![A-Za-z0-9_])%s\([^;\n]*?\);" % n, "", text)                    # ... or inside a line
    open(path, "w").write(text)
