"""moved next to the oracle (oracle/plan_decisions.py): __graft_entry__.smoke() uses it and must not depend on the test tree"""
from oracle.plan_decisions import *  # noqa: F401,F403
from oracle.plan_decisions import plan_decisions  # noqa: F401
