"""Segmentation scores on the GPU: confusion matrix -> per-class Dice / IoU.

Reference: `runningScore` (src/common_utils/metrics.py:12-52, confusion-matrix mean IoU used for validation) and
`runningMySegmentationScore` -> medpy.metric.binary.dc (metrics.py:216-218): Dice = 2|A n B| / (|A| + |B|), 0.0 when both are empty."""
import torch

from ._lib import lib, check


class runningScore:
    """Accumulates a K x K confusion matrix (rows = ground truth, columns = prediction) with a HIP kernel.  Same surface as the reference
    class (metrics.py:12-52: `update(label_trues, label_preds)`, `get_scores()` with its exact key strings, `reset()`); `update` also takes
    `logits=` [N,K,H,W] (argmax fused into the kernel).  The matrix is created on the device of the first update."""

    KEYS = ('Overall Acc: \t', 'Mean Acc : \t', 'FreqW Acc : \t', 'Mean IoU : \t')      # metrics.py:46-49, read by train_adv...py:87-88

    def __init__(self, n_classes, device=None):
        self.n_classes = n_classes
        self.cm = None if device is None else torch.zeros(n_classes * n_classes, dtype=torch.int64, device=device)

    def reset(self):
        if self.cm is not None:
            self.cm.zero_()

    def update(self, label_trues, label_preds=None, logits=None):
        if logits is None and torch.is_tensor(label_preds) and label_preds.dim() == 4 and label_preds.is_floating_point():
            logits, label_preds = label_preds, None          # [N,K,H,W] scores in the second position
        if logits is None:
            if label_preds is None:
                raise TypeError("update() needs label_preds (label maps) or logits")
            dev = self.cm.device if self.cm is not None else torch.device("cuda")
            lp = torch.as_tensor(label_preds).to(device=dev, dtype=torch.int64)
            if lp.dim() == 2:
                lp = lp.unsqueeze(0)
            logits = torch.zeros(lp.shape[0], self.n_classes, *lp.shape[1:], device=dev).scatter_(1, lp.clamp(0, self.n_classes - 1).unsqueeze(1), 1.0)
        if not logits.is_cuda:
            raise RuntimeError("maxstyle_amd.metrics runs on the MI355X only")
        logits = logits.contiguous().float()
        labels = torch.as_tensor(label_trues).to(device=logits.device, dtype=torch.int64).contiguous()
        N, K, H, W = logits.shape
        assert K == self.n_classes
        if self.cm is None:
            self.cm = torch.zeros(K * K, dtype=torch.int64, device=logits.device)
        check(lib.ms_confusion(logits.data_ptr(), labels.data_ptr(), self.cm.data_ptr(), N, K, H * W, torch.cuda.current_stream().cuda_stream), "ms_confusion")

    def confusion_matrix(self):
        if self.cm is None:
            return torch.zeros(self.n_classes, self.n_classes, dtype=torch.int64)
        return self.cm.view(self.n_classes, self.n_classes).clone()

    def get_scores(self):
        """metrics.py:29-49: ({overall acc, mean acc, freq-weighted acc, mean IoU} under the reference's key strings, per-class IoU)."""
        hist = self.confusion_matrix().double().cpu()
        acc = hist.diag().sum() / hist.sum()
        acc_cls = (hist.diag() / hist.sum(1)).nanmean()
        iu = hist.diag() / (hist.sum(1) + hist.sum(0) - hist.diag())
        freq = hist.sum(1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        k = self.KEYS
        return {k[0]: float(acc), k[1]: float(acc_cls), k[2]: float(fwavacc), k[3]: float(iu.nanmean())}, \
               dict(zip(range(self.n_classes), [float(v) for v in iu]))

    def dice(self):
        """Per foreground class 2|A n B| / (|A|+|B|) pooled over everything seen so far (0.0 when both are empty)."""
        hist = self.confusion_matrix().double()
        out = []
        for c in range(1, self.n_classes):
            denom = hist[c, :].sum() + hist[:, c].sum()
            out.append(0.0 if denom == 0 else float(2.0 * hist[c, c] / denom))
        return out
