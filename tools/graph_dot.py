"""Dev tool: node -> stream assignment of a captured step from the runtime's DOT dump.
On the GPU box:  DEBUG_HIP_GRAPH_DOT_PRINT=1 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-fk \
                 --no-kernel-timing --no-phase1      (writes graph_<pid>_dot_print_1 into the working directory)
then:            python tools/graph_dot.py graph_<pid>_dot_print_1
Prints every kernel node with the stream the runtime gave it and its successors in capture order.  What it showed (HIP 7.0.2 as
bundled with the torch wheel): a node's i-th captured successor runs on stream (node's stream + i) mod 4 -- the first one keeps
the stream, every other one starts behind a queue hand-over."""
import re, sys
s = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*label="(\d+)\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', s):
    nm = re.sub(r'^_Z\d+', '', re.sub(r'_ZN12_GLOBAL__N_1\d+', '', m.group(3)))
    nodes[int(m.group(1))] = (nm[:32], int(m.group(4)), m.group(5))
succ = {}
for a, b in re.findall(r'"graph_1_node_(\d+)"\s*->\s*"graph_1_node_(\d+)"', s):
    succ.setdefault(int(a), []).append(int(b))
for i in sorted(nodes):
    print("%3d  stream %d  %-32s -> %s" % (i, nodes[i][1], nodes[i][0], succ.get(i, [])))
