"""Representation metrics the reference's evaluation script calls (`from improved_diffusion import metrics as mt`,
scripts/image_causaldae_test.py:208,260,311 -> reference metrics.py:167-246).  Host-side numpy / scikit-learn evaluation code, not
part of the accelerated path; written from the published definitions:

  * DCI (Eastwood & Williams, "A framework for the quantitative evaluation of disentangled representations", ICLR 2018): one
    gradient-boosted regressor per generative factor on the codes; R[i, j] = importance of code i for factor j;
    disentanglement of code i = 1 - H(R[i, :] / sum) (entropy in base #factors), weighted by the code's share of importance;
    completeness of factor j = 1 - H(R[:, j] / sum) (base #codes), weighted by the factor's share.
  * MCC: mean absolute Pearson correlation under the best one-to-one matching of dimensions (Hungarian algorithm).

Array conventions follow the reference: codes `mus` are [num_codes, num_points], factors `ys` are [num_factors, num_points]."""
import numpy as np


def _entropy_rows(p, base):
    """Shannon entropy of each row of a non-negative matrix after normalising the row, in the given base."""
    p = p / p.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(p > 0, p * np.log(p), 0.0)
    return -t.sum(axis=1) / np.log(base)


def compute_importance_gbt(x_train, y_train, x_test, y_test):
    """Importance matrix [num_codes, num_factors] from one GradientBoostingRegressor per factor, and the train / test
    "informativeness" the reference reports (fraction of exactly reproduced targets, reference metrics.py:182-199)."""
    from sklearn.ensemble import GradientBoostingRegressor
    num_codes, num_factors = x_train.shape[0], y_train.shape[0]
    importance = np.zeros((num_codes, num_factors), dtype=np.float64)
    train_hit, test_hit = [], []
    for j in range(num_factors):
        model = GradientBoostingRegressor()
        model.fit(x_train.T, y_train[j])
        importance[:, j] = np.abs(model.feature_importances_)
        train_hit.append(np.mean(model.predict(x_train.T) == y_train[j]))
        test_hit.append(np.mean(model.predict(x_test.T) == y_test[j]))
    return importance, float(np.mean(train_hit)), float(np.mean(test_hit))


def disentanglement_per_code(importance_matrix):
    return 1.0 - _entropy_rows(importance_matrix + 1e-11, importance_matrix.shape[1])


def disentanglement(importance_matrix):
    """(score, per-code share of the total importance)."""
    per_code = disentanglement_per_code(importance_matrix)
    m = importance_matrix if importance_matrix.sum() != 0.0 else np.ones_like(importance_matrix)
    share = m.sum(axis=1) / m.sum()
    return float(np.sum(per_code * share)), share


def completeness_per_factor(importance_matrix):
    return 1.0 - _entropy_rows(importance_matrix.T + 1e-11, importance_matrix.shape[0])


def completeness(importance_matrix):
    per_factor = completeness_per_factor(importance_matrix)
    m = importance_matrix if importance_matrix.sum() != 0.0 else np.ones_like(importance_matrix)
    share = m.sum(axis=0) / m.sum()
    return float(np.sum(per_factor * share))


def _compute_dci(mus_train, ys_train, mus_test, ys_test):
    """-> (scores dict with informativeness_train/test, disentanglement, completeness; importance matrix; per-code importance share)."""
    importance, train_acc, test_acc = compute_importance_gbt(mus_train, ys_train, mus_test, ys_test)
    assert importance.shape == (mus_train.shape[0], ys_train.shape[0])
    disent, code_share = disentanglement(importance)
    scores = {"informativeness_train": train_acc, "informativeness_test": test_acc, "disentanglement": disent,
              "completeness": completeness(importance)}
    return scores, importance, code_share


def MCC(Z, Zp):
    """Mean correlation coefficient of two [num_points, n] representations under the best permutation of dimensions."""
    from scipy.optimize import linear_sum_assignment
    n = Z.shape[1]
    rho = np.abs(np.corrcoef(Z.T, Zp.T)[:n, n:])
    r, c = linear_sum_assignment(-rho)
    return float(rho[r, c].mean())
