"""CPU check of the valid-outcome rule the GPU edge-case tests judge by (tests/select_rule.py)."""
import torch

from tests.select_rule import feasible_classes, seed_rows, valid_victims


def test_valid_victims_rule_on_the_oracle_itself():
    """The membership rule accepts what torch itself decides on tied rows (CPU, any tie order) and rejects a wrong victim."""
    from oracle import easykv_oracle as O
    g = torch.Generator().manual_seed(5)
    for trial in range(40):
        H, W, budget = 3, 41, 40
        s, q, c, _ = seed_rows(H, W, 40, int(torch.randint(0, 25, (1,), generator=g)), g)
        c[:, -1] = 1.0
        ids = O._select_decode("roco", s, q, c, budget)
        std, mean = O.roco_std(s, q, c), s / c
        for h in range(H):
            forced, pool, need = feasible_classes(std[h], budget - int(budget * 0.3))
            assert valid_victims([int(ids[h])], mean[h], forced, pool, need, 1)
            worst = int(torch.argmax(torch.where(torch.isnan(mean[h]), torch.zeros(()), mean[h])))
            if worst != int(ids[h]):
                assert not valid_victims([worst], mean[h], forced, pool, need, 1)


