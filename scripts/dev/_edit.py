"""tiny helper for scripted edits: rep(path, old, new, count=1) asserts that `old` occurs exactly `count` times"""
def rep(path, old, new, count=1):
    s = open(path).read()
    n = s.count(old)
    assert n == count, "%s: %d occurrences of %r" % (path, n, old[:60])
    open(path, "w").write(s.replace(old, new))
