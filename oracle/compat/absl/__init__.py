"""Minimal stand-in for absl-py (TEST INFRASTRUCTURE ONLY; absl is not installed in this image).

Just enough for the reference's `example_run_loop.py` (`app.run`, `flags.DEFINE_*` / `FLAGS.<name>`,
`logging.info`) to be imported and its `main()` called from a test.  A real absl, if ever installed, wins:
oracle/ref_harness.py only appends this directory to sys.path when `import absl` fails."""
