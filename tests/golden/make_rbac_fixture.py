#!/usr/bin/env python3
"""Generates tests/golden/rbac_*.json by IMPORTING the reference's own RBAC generators.

Runs only in the authoring container (needs /root/reference); the reference never travels,
only the data this script emits does.  Generators used (pure Python: random + numpy):
  services/rbac_generator/tree_based_rbac_data_generator.py  (TreeBasedRBACDataGenerator)
  services/rbac_generator/random_rbac_data_generator.py      (RandomRBACDataGenerator)
Both draw from the unseeded global `random`; we seed it and record the seed.

Each fixture holds the generator's outputs (users, user_roles, permission pairs) plus, per
user, the document set the reference's row-level-security policy admits
(controller/baseline/pg_row_security/row_level_security.py:54-65), computed here with plain
Python sets from the generator's own role->documents output.

Usage:  python tests/golden/make_rbac_fixture.py
"""
import importlib.util
import json
import os
import random
import sys

import numpy as np

REF = "/root/reference/services/rbac_generator"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _visible(user_roles, perms, n_users):
    role_docs = {}
    for r, d in perms:
        role_docs.setdefault(int(r), set()).add(int(d))
    out = {}
    for u in range(1, n_users + 1):
        docs = set()
        for uu, r in user_roles:
            if uu == u:
                docs |= role_docs.get(int(r), set())
        out[str(u)] = sorted(docs)
    return out


def tree_fixture(seed, num_users, num_roles, num_docs, h, b0, b1):
    mod = _load("tree_based_rbac_data_generator")
    random.seed(seed)
    np.random.seed(seed)
    gen = mod.TreeBasedRBACDataGenerator(num_users=num_users, num_roles=num_roles,
                                         document_ids=range(1, num_docs + 1), h=h, b0=b0, b1=b1)
    users, user_roles, doc_assign, perms = gen.generate_rbac_data()
    user_roles = [[int(u), int(r)] for u, r in user_roles]
    perms = [[int(r), int(d)] for r, d in perms]
    return {
        "generator": "services/rbac_generator/tree_based_rbac_data_generator.py:TreeBasedRBACDataGenerator",
        "seed": seed,
        "params": {"num_users": num_users, "num_roles": num_roles, "num_docs": num_docs,
                   "h": h, "b0": b0, "b1": b1},
        "num_users": len(users),
        "user_roles": user_roles,
        "permissions": perms,
        "role_num_docs": {str(r): len(d) for r, d in doc_assign.items()},
        "visible_docs": _visible(user_roles, perms, num_users),
    }


def random_fixture(seed, num_users, num_roles, num_docs, m_roles, m_perms):
    mod = _load("random_rbac_data_generator")
    random.seed(seed)
    gen = mod.RandomRBACDataGenerator(num_users, num_roles, list(range(1, num_docs + 1)),
                                      m_roles, m_perms)
    users, roles, user_roles, perms = gen.generate_rbac_data()
    user_roles = [[int(u), int(r)] for u, r in user_roles]
    perms = [[int(r), int(d)] for r, d in perms]
    return {
        "generator": "services/rbac_generator/random_rbac_data_generator.py:RandomRBACDataGenerator",
        "seed": seed,
        "params": {"num_users": num_users, "num_roles": num_roles, "num_docs": num_docs,
                   "m_roles": m_roles, "m_perms": m_perms},
        "num_users": len(users),
        "user_roles": user_roles,
        "permissions": perms,
        "visible_docs": _visible(user_roles, perms, num_users),
    }


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are committed, nothing to do")
    fx = {
        # small tree: 60 users, 20 roles, 300 docs, same shape parameters as the reference default (h=4,b 3..4)
        "rbac_tree_small.json": tree_fixture(20251121, 60, 20, 300, 4, 3, 4),
        # multi-role users with overlapping roles (random generator; result dedup matters)
        "rbac_random_small.json": random_fixture(20251122, 40, 12, 300, 3, 60),
    }
    for name, data in fx.items():
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, "users", data["num_users"], "user_roles", len(data["user_roles"]),
              "perms", len(data["permissions"]))


if __name__ == "__main__":
    main()
