!> Re-creation of the reference's integration scenario (test/splpak_test.f90): 1-D noisy
!! fit, 10 nodes, 20 points, weights 1-|r|, xtrap = 1, fit error <= 1e-1 over 100 points.
!! The reference draws its noise from the compiler-specific random_number; a Park-Miller
!! stream (seed 42) is used here so the data are reproducible across compilers.
program test_noisy
    use splpak_module, wp => splpak_wp
    implicit none
    integer,parameter :: ndim = 1, nxdata = 20, nest = 100
    integer,dimension(ndim),parameter :: nodes = [10]
    integer,parameter :: ncol = product(nodes), nwrk = ncol*(ncol+1), ncf = ncol
    real(wp) :: xdata(ndim,nxdata), ydata(nxdata), wdata(nxdata), xmin(ndim), xmax(ndim)
    real(wp) :: work(nwrk), coef(ncf), x(ndim), f, tru, errmax, r
    integer(8) :: s
    integer :: i, ierror
    type(splpak_type) :: solver
    character(len=16) :: backend

    ! `<program> host`: the same scenario on the module's HOST solver (set_host; no GPU needed)
    call get_command_argument(1, backend)
    if (trim(backend) == 'host') call solver%set_host(.true.)

    xmin = 0.0_wp; xmax = 1.0_wp
    s = 42_8
    do i = 1, nxdata
        s = mod(48271_8*s, 2147483647_8)
        r = (real(s,wp)/2147483647.0_wp - 0.5_wp)*0.1_wp        ! noise in +-0.05
        xdata(1,i) = real(i-1,wp)/real(nxdata-1,wp)
        ydata(i) = f1(xdata(1,i)) + r
        wdata(i) = 1.0_wp - abs(r)
    end do
    call solver%initialize(ndim,xdata,ndim,ydata,wdata,nxdata,xmin,xmax,nodes,1.0_wp,coef,ncf,work,nwrk,ierror)
    write(*,*) 'splcw ierror = ', ierror
    if (ierror /= 0) error stop 'error calling splcw'
    errmax = 0.0_wp
    do i = 1, nest
        x(1) = real(i-1,wp)/nest
        f = solver%evaluate(ndim,x,coef,xmin,xmax,nodes,ierror)
        if (ierror /= 0) error stop 'error calling splfe'
        tru = f1(x(1))
        errmax = max(errmax, abs(tru - f))
    end do
    write(*,*) 'splfe errmax = ', errmax
    if (errmax > 1.0e-1_wp) error stop 'errmax too large'
    write(*,*) 'PASS test_noisy'
contains
    real(wp) function f1(x)
        real(wp),intent(in) :: x
        f1 = 0.5_wp*(x*exp(-x) + sin(x))
    end function f1
end program test_noisy
