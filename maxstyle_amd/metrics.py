"""Segmentation scores on the GPU: confusion matrix -> per-class Dice / IoU.

Reference: `runningScore` (src/common_utils/metrics.py:12-52, confusion-matrix mean IoU used for validation) and
`runningMySegmentationScore` -> medpy.metric.binary.dc (metrics.py:216-218): Dice = 2|A n B| / (|A| + |B|), 0.0 when both are empty."""
import torch

from ._lib import lib, check


class runningScore:
    """Accumulates a K x K confusion matrix (rows = ground truth, columns = prediction) with a HIP kernel."""

    def __init__(self, n_classes, device="cuda"):
        self.n_classes = n_classes
        self.cm = torch.zeros(n_classes * n_classes, dtype=torch.int64, device=device)

    def reset(self):
        self.cm.zero_()

    def update(self, label_trues, logits):
        if not logits.is_cuda:
            raise RuntimeError("maxstyle_amd.metrics runs on the MI355X only")
        logits = logits.contiguous().float()
        labels = label_trues.to(device=logits.device, dtype=torch.int64).contiguous()
        N, K, H, W = logits.shape
        assert K == self.n_classes
        check(lib.ms_confusion(logits.data_ptr(), labels.data_ptr(), self.cm.data_ptr(), N, K, H * W, torch.cuda.current_stream().cuda_stream), "ms_confusion")

    def confusion_matrix(self):
        return self.cm.view(self.n_classes, self.n_classes).clone()

    def get_scores(self):
        """Same quantities as metrics.py:30-52: overall acc, mean acc, freq-weighted acc, mean IoU, per-class IoU."""
        hist = self.confusion_matrix().double()
        acc = hist.diag().sum() / hist.sum()
        acc_cls = (hist.diag() / hist.sum(1)).nanmean()
        iu = hist.diag() / (hist.sum(1) + hist.sum(0) - hist.diag())
        freq = hist.sum(1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        return {"Overall Acc": float(acc), "Mean Acc": float(acc_cls), "FreqW Acc": float(fwavacc), "Mean IoU": float(iu.nanmean())}, \
               dict(zip(range(self.n_classes), [float(v) for v in iu]))

    def dice(self):
        """Per foreground class 2|A n B| / (|A|+|B|) pooled over everything seen so far (0.0 when both are empty)."""
        hist = self.confusion_matrix().double()
        out = []
        for c in range(1, self.n_classes):
            denom = hist[c, :].sum() + hist[:, c].sum()
            out.append(0.0 if denom == 0 else float(2.0 * hist[c, c] / denom))
        return out
