"""Newick -> the reference's nlist arrays.

tree_build_node_list (cafe/cafe_commands.cpp:2028-2051) numbers nodes by in-order traversal, so
even ids are leaves and odd ids internal; the family table's column for leaf slot j belongs to
node 2j (cafe/gene_family.cpp:413-445 matches species to leaves by case-insensitive name).
"""
import numpy as np


class CafeTree:
    def __init__(self, newick):
        s = newick.strip().rstrip(";")
        names, bls, kids = [], [], []
        pos = 0

        def parse():
            nonlocal pos
            me = len(names)
            names.append("")
            bls.append(-1.0)  # the root keeps branchlength -1 (libtree/phylogeny.c)
            kids.append([])
            if s[pos] == "(":
                pos += 1
                while True:
                    kids[me].append(parse())
                    if s[pos] == ",":
                        pos += 1
                        continue
                    if s[pos] != ")":
                        raise ValueError("malformed Newick at %d" % pos)
                    pos += 1
                    break
            j = pos
            while j < len(s) and s[j] not in ",():":
                j += 1
            names[me] = s[pos:j]
            pos = j
            if pos < len(s) and s[pos] == ":":
                k = pos + 1
                while k < len(s) and s[k] not in ",()":
                    k += 1
                bls[me] = float(s[pos + 1:k])
                pos = k
            return me

        root = parse()
        order = []
        stack = [(root, 0)]
        while stack:
            v, st = stack.pop()
            if not kids[v]:
                order.append(v)
            elif st == 0:
                if len(kids[v]) != 2:
                    raise ValueError("Tree must be binary")
                stack.append((v, 1))
                stack.append((kids[v][0], 0))
            else:
                order.append(v)
                stack.append((kids[v][1], 0))
        new = {old: i for i, old in enumerate(order)}
        n = len(order)
        self.n_nodes = n
        self.parent = np.full(n, -1, np.int32)
        self.left = np.full(n, -1, np.int32)
        self.right = np.full(n, -1, np.int32)
        self.branchlength = np.full(n, -1.0)
        self.name = [""] * n
        for old in range(n):
            i = new[old]
            self.name[i] = names[old]
            self.branchlength[i] = bls[old]
            if kids[old]:
                a, b = new[kids[old][0]], new[kids[old][1]]
                self.left[i], self.right[i] = a, b
                self.parent[a] = self.parent[b] = i
        self.root = new[root]
        self.n_leaves = (n + 1) // 2
        self.leaf_names = [self.name[i] for i in range(0, n, 2)]

    def column_order(self, species):
        """Columns of a family table (species order) -> leaf-slot order."""
        low = [x.lower() for x in species]
        return [low.index(nm.lower()) for nm in self.leaf_names]

    def max_branch_length(self):
        return float(self.branchlength.max())

    def apply(self, engine):
        engine.set_tree(self.parent, self.left, self.right, self.branchlength)
