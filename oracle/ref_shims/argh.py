"""TEST INFRASTRUCTURE ONLY -- stand-in for argh (kindel/cli.py:2,63-66); the reference
CLI is never driven through this shim, it only has to import."""


class ArghParser:
    def add_commands(self, fns):
        self.fns = fns

    def dispatch(self):
        raise SystemExit("argh shim: CLI dispatch not supported")
