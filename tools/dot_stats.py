"""Shape of a hipGraphDebugDotPrint dump: nodes, edges, critical path length, level widths."""
import re, sys, collections
txt = open(sys.argv[1]).read()
edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
nodes = set(re.findall(r'^\s*"?([\w.]+)"?\s*\[', txt, flags=re.M))
for a, b in edges:
    nodes.add(a); nodes.add(b)
succ = collections.defaultdict(list); indeg = collections.Counter()
for a, b in edges:
    succ[a].append(b); indeg[b] += 1
level = {}
q = collections.deque(n for n in nodes if indeg[n] == 0)
for n in q: level[n] = 0
ind = dict(indeg)
while q:
    n = q.popleft()
    for m in succ[n]:
        level[m] = max(level.get(m, 0), level[n] + 1)
        ind[m] -= 1
        if ind[m] == 0: q.append(m)
depth = max(level.values()) + 1 if level else 0
width = collections.Counter(level.values())
print("nodes %d edges %d depth %d  (nodes/depth = %.2f)" % (len(nodes), len(edges), depth, len(nodes) / max(depth, 1)))
print("levels with width>1: %d ; max width %d" % (sum(1 for w in width.values() if w > 1), max(width.values())))
multi_out = sum(1 for n in nodes if len(succ[n]) > 1); multi_in = sum(1 for n in nodes if indeg[n] > 1)
print("fork nodes %d, join nodes %d" % (multi_out, multi_in))
