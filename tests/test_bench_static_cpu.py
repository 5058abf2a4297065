"""A static check of bench.py's multi-rank control flow (CPU): every torch.distributed collective must be reached by EVERY
rank. Round 3 broke that once — rank 0's clock ramp in front of the side measurements called the sharded step(), whose
all-reduce the other ranks (already waiting at the final barrier) never joined, and every run with more than one rank hung.
The GPU suite has the dynamic test (tests/test_gpu_bench_contract.py: two ranks through torch.distributed.run on one GPU);
this one runs wherever the repository is checked out."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLLECTIVES = {"all_reduce", "barrier", "broadcast", "all_gather", "all_gather_into_tensor", "reduce_scatter", "all_to_all",
               "gather", "scatter", "reduce", "new_group", "init_process_group", "destroy_process_group"}


def _is_dist_call(node):
    return (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in COLLECTIVES and
            isinstance(node.func.value, ast.Name) and node.func.value.id == "dist")


def _local_functions(fn):
    return {n.name: n for n in ast.walk(fn) if isinstance(n, (ast.FunctionDef, ast.Lambda)) and hasattr(n, "name")}


def _has_collective(node, funcs, seen=()):
    """Does this subtree issue a collective, directly or through a function defined in main() / at module level?"""
    for n in ast.walk(node):
        if _is_dist_call(n):
            return True
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in funcs and n.func.id not in seen:
            if _has_collective(funcs[n.func.id], funcs, seen + (n.func.id,)):
                return True
        # a function handed on as a value (keep_gpu_warm(lambda: step(...)), keep_gpu_warm(local_step))
        if isinstance(n, ast.Name) and n.id in funcs and n.id not in seen and isinstance(n.ctx, ast.Load):
            if _has_collective(funcs[n.id], funcs, seen + (n.id,)):
                return True
    return False


def _rank_zero_blocks(fn):
    for n in ast.walk(fn):
        if isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and isinstance(n.test.left, ast.Name) and \
                n.test.left.id == "rank" and len(n.test.ops) == 1 and isinstance(n.test.ops[0], ast.Eq) and \
                isinstance(n.test.comparators[0], ast.Constant) and n.test.comparators[0].value == 0:
            yield n


def test_no_collective_is_reachable_from_a_rank_zero_only_block():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    module_funcs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    main = module_funcs["main"]
    funcs = dict(module_funcs)
    funcs.update({n.name: n for n in ast.walk(main) if isinstance(n, ast.FunctionDef) and n is not main})
    # the checker sees what it is meant to see: step() and timed() do issue collectives, rewarm() must not
    assert _has_collective(funcs["step"], funcs) and _has_collective(funcs["timed"], funcs)
    assert not _has_collective(funcs["rewarm"], funcs)
    blocks = list(_rank_zero_blocks(main))
    assert len(blocks) >= 2
    for b in blocks:
        for stmt in b.body:     # (the else-branch, if any, is the other ranks' business)
            assert not _has_collective(stmt, funcs), "collective reachable from `if rank == 0:` at bench.py:%d" % stmt.lineno


def test_the_checker_catches_the_round_3_bug():
    src = '''
def main():
    def step(sh):
        dist.all_reduce(sh)
    def rewarm():
        keep_gpu_warm(lambda: step(1), 0.1)
    if rank == 0:
        rewarm()
'''
    tree = ast.parse(src)
    main = tree.body[0]
    funcs = {n.name: n for n in ast.walk(main) if isinstance(n, ast.FunctionDef) and n is not main}
    blocks = list(_rank_zero_blocks(main))
    assert len(blocks) == 1 and _has_collective(blocks[0].body[0], funcs)
