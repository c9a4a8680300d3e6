"""Logger file formats: progress.csv with a growing header, the dashed log.txt table, progress.json lines,
running means — the formats of improved_diffusion/logger.py:36-143,350-353."""
import json

import numpy as np

from causaldiffae_amd import logger


def test_csv_header_growth_and_table(tmp_path, capsys):
    logger.configure(dir=str(tmp_path), format_strs=["stdout", "log", "csv", "json"])
    assert logger.get_dir() == str(tmp_path)
    logger.logkv("step", 0)
    logger.logkv_mean("loss", 1.0)
    logger.logkv_mean("loss", 2.0)
    logger.logkv_mean("loss", 6.0)
    out = logger.dumpkvs()
    assert out == {"step": 0, "loss": 3.0}
    logger.logkv("step", 1)
    logger.logkv("a_new_key", np.float32(0.5))
    logger.logkv("zeta", "text")
    logger.dumpkvs()
    logger.logkv("step", 2)
    logger.dumpkvs()
    logger.log("free", "text", 3)
    logger.reset()
    rows = (tmp_path / "progress.csv").read_text().splitlines()
    assert rows[0] == "loss,step,a_new_key,zeta"          # first keys sorted, later keys sorted and appended
    assert rows[1] == "3.0,0,,"                           # old row padded
    assert rows[2] == ",1,0.5,text"
    assert rows[3] == ",2,,"
    js = [json.loads(l) for l in (tmp_path / "progress.json").read_text().splitlines()]
    assert js[0] == {"loss": 3.0, "step": 0} and js[1]["a_new_key"] == 0.5 and len(js) == 3
    txt = (tmp_path / "log.txt").read_text().splitlines()
    assert txt[0].startswith("Logging to ")
    assert txt[1] == "-" * len(txt[1]) and txt[2] == "| loss | 3        |" and txt[3] == "| step | 0        |" and txt[4] == txt[1]
    assert "free text 3" in txt
    assert "| a_new_key | 0.5      |" in txt
    assert "| loss | 3        |" in capsys.readouterr().out


def test_long_keys_truncate_and_rank_suffix(tmp_path, monkeypatch):
    monkeypatch.setenv("RANK", "2")
    logger.configure(dir=str(tmp_path))
    logger.logkv("k" * 40, 1.23456789)
    logger.dumpkvs()
    logger.reset()
    assert not (tmp_path / "progress-rank002.csv").exists()          # non-zero ranks default to "log" only
    txt = (tmp_path / "log-rank002.txt").read_text()
    assert "| " + "k" * 27 + "... | 1.23     |" in txt


def test_log_loss_dict_quartiles(tmp_path):
    import torch
    from causaldiffae_amd.train_util import log_loss_dict

    class D:
        num_timesteps = 1000

    logger.configure(dir=str(tmp_path), format_strs=["csv"])
    t = torch.tensor([0, 249, 250, 999])
    log_loss_dict(D, t, {"loss": torch.tensor([1.0, 3.0, 5.0, 7.0]), "kld_rep": torch.tensor(2.0)})
    kv = logger.dumpkvs()
    logger.reset()
    assert kv["loss"] == 4.0 and kv["loss_q0"] == 2.0 and kv["loss_q1"] == 5.0 and kv["loss_q3"] == 7.0 and "loss_q2" not in kv
    assert kv["kld_rep"] == 2.0 and not any(k.startswith("kld_rep_q") for k in kv)
