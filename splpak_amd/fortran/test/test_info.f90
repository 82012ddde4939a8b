!> Diagnostics of a fit through the Fortran drop-in: `last_fit_info` returns the residual norm the
!! reference computes (suprls, src/splpak.F90:1693) and drops (splcw :690), the row counts and the
!! measured optimality residual.  The data are parity case "2d16" of tests/cases.py (2-D, 16x16 nodes,
!! 10^4 weighted points of the seeded Park-Miller stream of SURVEY 8d, xtrap = 1), so that the Python
!! test can compare the printed `reserr` with the oracle's value for the same inputs.  Also: the
!! scalar (host) and the batched (GPU) evaluation agree, with and without derivatives.
program test_info
    use splpak_module, wp => splpak_wp
    implicit none
    integer,parameter :: m = 10000, nq = 500
    integer :: nodes(2), ierror, i, nrows, ncons, nsteps, nbad
    real(wp) :: xdata(2,m), ydata(m), wdata(m), xmin(2), xmax(2), coef(256), work(256*257)
    real(wp) :: xs(2,nq), fs(nq), f, reserr, omega, u(4), coef2(256)
    character(len=16) :: envval
    integer(8) :: s
    type(splpak_type) :: solver

    nbad = 0
    s = 42_8
    do i = 1, m
        call draw(u(1)); call draw(u(2)); call draw(u(3)); call draw(u(4))
        xdata(1,i) = u(1)
        xdata(2,i) = u(2)
        ydata(i) = sin(3.0_wp*u(1) + 1.0_wp) + sin(3.0_wp*u(2) + 2.0_wp) + 0.01_wp*(u(3) - 0.5_wp)
        wdata(i) = 0.5_wp + u(4)
    end do
    do i = 1, nq
        call draw(u(1)); call draw(u(2))
        xs(1,i) = 1.5_wp*u(1) - 0.25_wp            ! also outside the grid
        xs(2,i) = 1.5_wp*u(2) - 0.25_wp
    end do
    xmin = 0.0_wp; xmax = 1.0_wp; nodes = [16,16]
    call solver%initialize(2,xdata,2,ydata,wdata,m,xmin,xmax,nodes,1.0_wp,coef,256,work,256*257,ierror)
    if (ierror /= 0) error stop 'fit failed'
    call solver%last_fit_info(reserr=reserr, ndata_rows=nrows, nconstraint_rows=ncons, refine_steps=nsteps, &
                              optimality=omega)
    write(*,'(A,ES24.16)') ' reserr = ', reserr
    write(*,'(A,I8,I8,I4,ES12.3)') ' rows, constraint rows, steps, optimality = ', nrows, ncons, nsteps, omega
    if (nrows /= m) call fail('data row count')
    if (ncons < 1) call fail('constraint row count')
    if (.not. (omega < 1.0e-12_wp)) call fail('optimality residual')
    if (.not. (reserr > 0.0_wp)) call fail('reserr')

    call solver%evaluate_many(2,nq,xs,2,coef,xmin,xmax,nodes,fs,ierror)
    if (ierror /= 0) error stop 'evaluate_many failed'
    do i = 1, nq
        f = solver%evaluate(2,xs(:,i),coef,xmin,xmax,nodes,ierror)
        if (abs(f - fs(i)) > 1.0e-13_wp*max(1.0_wp,abs(f))) call fail('scalar vs batched value')
    end do
    call solver%evaluate_many(2,nq,xs,2,[1,2],coef,xmin,xmax,nodes,fs,ierror)
    do i = 1, nq
        f = solver%evaluate(2,xs(:,i),[1,2],coef,xmin,xmax,nodes,ierror)
        if (abs(f - fs(i)) > 1.0e-12_wp*max(3375.0_wp,abs(f))) call fail('scalar vs batched derivative')
    end do
    ! the same fit spread over 2 GPUs (SPLPAK_VIRTUAL_GPUS=1 in the environment: 2 ranks on this GPU)
    call get_environment_variable('SPLPAK_VIRTUAL_GPUS', envval, status=i)
    if (i == 0) then
        coef2 = coef
        call solver%set_gpus(2)
        call solver%initialize(2,xdata,2,ydata,wdata,m,xmin,xmax,nodes,1.0_wp,coef,256,work,256*257,ierror)
        if (ierror /= 0) error stop 'multi-GPU fit failed'
        if (maxval(abs(coef - coef2)) > 1.0e-12_wp*maxval(abs(coef2))) call fail('2-GPU fit differs from 1-GPU fit')
        call solver%last_fit_info(ndata_rows=nrows)
        if (nrows /= m) call fail('data row count over the shards')
        write(*,'(A,ES12.3)') ' 2-GPU vs 1-GPU coefficients: ', maxval(abs(coef - coef2))/maxval(abs(coef2))
        call solver%set_gpus(1)
    end if
    call solver%destroy()
    if (nbad /= 0) error stop 'test_info FAILED'
    write(*,*) 'PASS test_info'
contains
    subroutine draw(v)
        real(wp),intent(out) :: v
        s = mod(48271_8*s, 2147483647_8)
        v = real(real(s,8)/2147483647.0_8, wp)
    end subroutine draw
    subroutine fail(what)
        character(len=*),intent(in) :: what
        write(*,*) 'FAILED: ', what
        nbad = nbad + 1
    end subroutine fail
end program test_info
