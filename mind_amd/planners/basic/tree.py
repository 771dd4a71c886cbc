"""Keyed tree container with the call surface of the reference ``planners/basic/tree.py:1-110``
(``Node(key, parent_key, data)``, ``Tree.add_node / get_node / get_root / get_leaf_nodes /
retrieve_nodes_to_root / size / leaves / nodes``), which ``common/visualization.py:218-281`` and
``agent.py`` rely on for the returned scenario / trajectory trees.

Differences in implementation only: leaves are kept in an insertion-ordered dict (O(1) parent
removal instead of list.remove), same observable order.
"""


class Node:
    __slots__ = ("key", "parent_key", "children_keys", "data", "depth")

    def __init__(self, key, parent_key, data):
        self.key = key
        self.parent_key = parent_key
        self.children_keys = []
        self.data = data
        self.depth = 0

    def __str__(self):
        return f"Node_{self.key}: Parent: {self.parent_key}, Children: {self.children_keys}"


class Tree:
    def __init__(self):
        self.nodes = {}
        self.root = None
        self._leaves = {}

    # -- lookups -------------------------------------------------------------------------------
    def _need(self, key):
        try:
            return self.nodes[key]
        except KeyError:
            raise KeyError("Node with the given key does not exist.") from None

    def get_node(self, key):
        return self._need(key)

    def get_parent_key(self, key):
        return self.nodes[self._need(key).parent_key]

    def has_children(self, key):
        return len(self._need(key).children_keys) > 0

    def get_children_keys(self, key):
        return self._need(key).children_keys

    def get_root(self):
        if self.root is None:
            raise KeyError("root node does not exist.")
        return self.nodes[self.root]

    def get_root_key(self):
        if self.root is None:
            raise KeyError("root node does not exist.")
        return self.root

    @property
    def leaves(self):
        return list(self._leaves)

    def get_leaf_keys(self):
        return self.leaves

    def get_leaf_nodes(self):
        return [self.nodes[k] for k in self._leaves]

    def size(self):
        return len(self.nodes)

    # -- mutation ------------------------------------------------------------------------------
    def add_node(self, node):
        if node.parent_key is None and not self.nodes:
            self.nodes[node.key] = node
            self.root = node.key
            self._leaves[node.key] = None
            return
        if node.parent_key not in self.nodes:
            raise KeyError("Parent does not exist.")
        if node.key in self.nodes:
            raise ValueError("Node key already exists.")
        parent = self.nodes[node.parent_key]
        parent.children_keys.append(node.key)
        self._leaves.pop(node.parent_key, None)
        node.depth = parent.depth + 1
        self.nodes[node.key] = node
        self._leaves[node.key] = None

    # -- traversal -----------------------------------------------------------------------------
    def retrieve_nodes_to_root(self, key):
        out = [self._need(key)]
        while out[-1].parent_key is not None:
            out.append(self._need(out[-1].parent_key))
        return out

    def process_up_down(self, fcn):
        if self.root is None:
            raise KeyError("root node does not exist.")
        stack = [self.root]
        while stack:
            n = self.nodes[stack.pop()]
            fcn(n)
            stack.extend(reversed(n.children_keys))

    def print(self):
        self.process_up_down(print)


class LazyTree(Tree):
    """A Tree whose nodes are built on first access.  The native planner returns trees of tens of thousands of nodes whose hot-path
    consumers (the contingency solver, the candidate evaluation) read the flattened arrays that come with them and never the Python
    objects; everything that does look (``nodes``, ``leaves``, ``get_node`` ...: the visualiser, tests, the round-by-round code) sees an
    ordinary Tree.  ``builder(tree)`` fills the tree through ``add_node``; ``root`` may be known before."""

    def __init__(self, builder, root=None):
        self._builder = builder
        self._nodes, self._lv = {}, {}
        self.root = root

    def _materialise(self):
        b, self._builder = self._builder, None
        if b is not None:
            self.root = None          # (add_node sets it again with the first node)
            b(self)

    @property
    def nodes(self):
        self._materialise()
        return self._nodes

    @nodes.setter
    def nodes(self, v):
        self._nodes = v

    @property
    def _leaves(self):
        self._materialise()
        return self._lv

    @_leaves.setter
    def _leaves(self, v):
        self._lv = v

    def get_root_key(self):
        if self.root is None:
            self._materialise()
        return super().get_root_key()

    def get_root(self):
        self._materialise()
        return super().get_root()
