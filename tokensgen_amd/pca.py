"""PCA holder for the T2To tail.  The reference pickles a whole `pca.PCA` module (pca.py:6-66) and `torch.load`s it in the
pipeline (pipeline_cogvideox_t2to.py:771); `compat.install_longvgen_alias()` registers this class under the module name
`pca` so such a pickle unpickles here.  Only the buffers matter to the hot path: mean_ [1, D] and components_ [d, D]."""
import torch
import torch.nn as nn


class PCA(nn.Module):
    def __init__(self, n_components=None):
        super().__init__()
        self.n_components = n_components

    @torch.no_grad()
    def fit(self, X):
        """SVD of the centred data, signs fixed by the largest-|u| entry of every left vector (pca.py:11-51)."""
        d = X.shape[1] if self.n_components is None else min(self.n_components, X.shape[1])
        self.register_buffer("mean_", X.mean(0, keepdim=True))
        U, _, Vt = torch.linalg.svd(X - self.mean_, full_matrices=False)
        signs = torch.sign(U[torch.argmax(U.abs(), dim=0), torch.arange(U.shape[1])])
        self.register_buffer("components_", (Vt * signs[:, None])[:d])
        return self

    def transform(self, X):
        return torch.matmul(X - self.mean_, self.components_.t())

    forward = transform

    def fit_transform(self, X):
        return self.fit(X).transform(X)

    def inverse_transform(self, Y):
        return torch.matmul(Y, self.components_) + self.mean_
