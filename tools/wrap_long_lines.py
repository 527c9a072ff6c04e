"""Re-wrap physical lines longer than LIMIT columns in Python sources WITHOUT changing what they mean: the result must parse to the
same AST (checked; a file whose AST would change is left untouched).  Breaks are only made inside brackets -- after a comma or before a
binary operator at the outermost possible depth -- or inside a string literal that is itself inside brackets (split at a space into two
adjacent literals), and comments are re-flowed.  Usage: python3 tools/wrap_long_lines.py [--limit 140] file.py ..."""
import ast
import io
import sys
import tokenize

LIMIT = 140


def _string_prefix(tok):
    q = min(p for p in (tok.find("'"), tok.find('"')) if p >= 0)
    return tok[:q], tok[q:]


def _split_string(tok, room):
    """a one-line string literal -> (head literal, tail literal) with the head at most `room` columns, or None"""
    prefix, body = _string_prefix(tok)
    if body[:3] in ('"""', "'''") or "r" in prefix.lower() and "\\" in body:
        return None
    quote = body[0]
    inner = body[1:-1]
    budget = room - len(prefix) - 2
    if budget < 20 or len(inner) <= budget:
        return None
    cut = inner.rfind(" ", 0, budget)
    while cut > 0:
        head = inner[:cut + 1]
        # never cut inside an escape or inside the braces of an f-string field
        depth = 0
        if "f" in prefix.lower():
            q = 0
            while q < len(head):
                if head[q] == "{":
                    if head[q:q + 2] == "{{":
                        q += 2
                        continue
                    depth += 1
                elif head[q] == "}":
                    if head[q:q + 2] == "}}" and depth == 0:
                        q += 2
                        continue
                    depth -= 1
                q += 1
        if depth == 0 and not head.endswith("\\") and head.count("\\") == 0 or depth == 0 and not head.rstrip(" ").endswith("\\"):
            return prefix + quote + head + quote, prefix + quote + inner[cut + 1:] + quote
        cut = inner.rfind(" ", 0, cut)
    return None


def _wrap_comment(line, limit):
    indent = len(line) - len(line.lstrip())
    text = line.strip()
    if not text.startswith("#"):
        return None
    words, out, cur = text[1:].split(), [], "#"
    for w in words:
        if len(cur) + 1 + len(w) + indent > limit and cur != "#":
            out.append(" " * indent + cur)
            cur = "#"
        cur += " " + w
    out.append(" " * indent + cur)
    return out


def _break_line(line, limit, in_brackets_at_start):
    """One physical line that is too long -> two physical lines (the second may still be too long: the caller iterates), or None."""
    stripped = line.lstrip()
    indent = len(line) - len(stripped)
    if stripped.startswith("#"):
        return _wrap_comment(line, limit)
    toks = []
    try:      # a physical line of a longer statement ends with open brackets: keep the tokens that came before the tokenizer gave up
        for t in tokenize.generate_tokens(io.StringIO(line + "\n").readline):
            toks.append(t)
    except (tokenize.TokenError, IndentationError, SyntaxError):
        pass
    if not toks:
        return None
    depth = 1 if in_brackets_at_start else 0
    stack_cols = []
    best = None          # (priority depth, column to break at, continuation indent)
    for t in toks:
        if t.type == tokenize.OP and t.string in "([{":
            depth += 1
            stack_cols.append(t.end[1])
        elif t.type == tokenize.OP and t.string in ")]}":
            depth -= 1
            if stack_cols:
                stack_cols.pop()
        elif t.type == tokenize.OP and t.string == "," and depth >= 1 and t.end[1] < limit - 1:
            cont = stack_cols[-1] if stack_cols else (indent if in_brackets_at_start else indent + 4)   # a continuation line keeps its column
            cand = (depth, t.end[1], cont)
            if best is None or cand[0] < best[0] or (cand[0] == best[0] and cand[1] > best[1]) or (best[1] < limit // 2 and cand[1] > best[1]):
                best = cand
        elif t.type == tokenize.COMMENT and depth == 0 and t.start[1] > 0 and len(line) > limit:
            # a trailing comment that makes the line too long goes above the statement
            code = line[:t.start[1]].rstrip()
            if len(code) <= limit and code.strip():
                return (_wrap_comment(" " * indent + t.string, limit) or []) + [code]
    if best is not None and best[1] > indent + 20:
        head, tail = line[:best[1]].rstrip(), line[best[1]:].strip()
        if tail:
            cont = min(best[2], limit // 2)
            return [head, " " * cont + tail]
    # a long string literal inside brackets
    depth = 1 if in_brackets_at_start else 0
    stack_cols = []
    for t in toks:
        if t.type == tokenize.OP and t.string in "([{":
            depth += 1
            stack_cols.append(t.end[1])
        elif t.type == tokenize.OP and t.string in ")]}":
            depth -= 1
            if stack_cols:
                stack_cols.pop()
        elif t.type == tokenize.STRING and depth >= 1 and t.end[1] > limit and t.start[0] == t.end[0]:
            parts = _split_string(t.string, limit - t.start[1])
            if parts:
                return [line[:t.start[1]] + parts[0], " " * t.start[1] + parts[1] + line[t.end[1]:]]
    return None


def wrap_source(src, limit=LIMIT):
    lines = src.split("\n")
    # bracket depth at the start of every physical line
    depth_at = [0] * (len(lines) + 1)
    in_string = set()        # physical lines (0-based) that belong to a string literal spanning several lines: text, not code
    try:
        depth = 0
        for t in tokenize.generate_tokens(io.StringIO(src).readline):
            if t.type == tokenize.OP and t.string in "([{":
                depth += 1
            elif t.type == tokenize.OP and t.string in ")]}":
                depth -= 1
            if t.type == tokenize.STRING and t.end[0] > t.start[0]:
                in_string.update(range(t.start[0] - 1, t.end[0]))
            if t.type in (tokenize.NL, tokenize.NEWLINE):
                depth_at[t.end[0]] = depth
    except tokenize.TokenError:
        return src
    out = []
    for no, line in enumerate(lines):
        if no in in_string:
            out.append(line)
            continue
        inside = depth_at[no] > 0 if no < len(depth_at) else False
        work = [line]
        for _ in range(12):
            longest = max(range(len(work)), key=lambda q: len(work[q]))
            if len(work[longest]) <= limit:
                break
            broken = _break_line(work[longest], limit, inside or longest > 0)
            if not broken:
                break
            work[longest:longest + 1] = broken
        out.extend(work)
    return "\n".join(out)


def main(argv):
    limit = LIMIT
    if argv and argv[0] == "--limit":
        limit = int(argv[1])
        argv = argv[2:]
    for path in argv:
        src = open(path).read()
        new = wrap_source(src, limit)
        try:
            same = ast.dump(ast.parse(src)) == ast.dump(ast.parse(new))
        except SyntaxError as e:
            print(f"{path}: NOT changed (result does not parse: {e})")
            continue
        if not same:
            print(f"{path}: NOT changed (the AST would differ)")
            continue
        left = sum(len(ln) > limit for ln in new.split("\n"))
        if new != src:
            open(path, "w").write(new)
        print(f"{path}: {sum(len(ln) > limit for ln in src.split(chr(10)))} long lines -> {left}")


if __name__ == "__main__":
    main(sys.argv[1:])
