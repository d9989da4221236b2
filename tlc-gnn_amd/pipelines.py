"""Drop-in for the forward half of the reference's pipelines.py (the TLC-GNN link-prediction harness).

  train :10-18 (forward + loss; the optimiser step is out of scope), test :20-40, weights_init :42-46, setup_seed :49-53.

The reference's `train()` / `test()` are closures over the module-level `model`, `data`, `optimizer`; here they take them as
arguments.  Everything between the arguments and the returned numbers is the reference's: encode once, decode per split,
binary cross-entropy, roc_auc_score / average_precision_score on the host.  The forward itself runs on the HIP kernels behind
baselines/TLCGNN.py (Net.encode / Net.decode)."""
import numpy as np
import torch
import torch.nn.functional as F


def weights_init(m):
    """:42-46  xavier_normal_ on every nn.Linear weight, zero bias (`model.apply(weights_init)`, :108)."""
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)


def setup_seed(seed):
    """:49-53"""
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


def train_forward(model, data):
    """The forward of train() (:11-15): model.train(), encode, decode('train') with its np.random.randint negatives, BCE.
    Returns (x, y, loss); loss.backward() / optimizer.step() (:16-17) are training and not provided.
    Training mode draws dropout masks, which the HIP encoder applies through torch's RNG (F.dropout in Net.encode)."""
    model.train()
    with torch.no_grad():
        emb = model.encode(data)
        x, y = model.decode(data, emb)
        loss = F.binary_cross_entropy(x, y)
    return x, y, loss


def test(model, data):
    """:20-40 -> [val BCE, val ROC-AUC, val AP, test ROC-AUC, test AP]."""
    from sklearn.metrics import roc_auc_score, average_precision_score
    model.eval()
    accs = []
    with torch.no_grad():
        emb = model.encode(data)
        for split in ["val", "test"]:
            pred, y = model.decode(data, emb, type=split)
            pred, y = pred.cpu(), y.cpu()
            if split == "val":
                accs.append(F.binary_cross_entropy(pred, y))
            pred = pred.data.numpy()
            accs.append(roc_auc_score(y, pred))
            accs.append(average_precision_score(y, pred))
    return accs
